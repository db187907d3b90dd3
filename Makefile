# Convenience targets; the authoritative build entry is __graft_entry__.build().
PY ?= python

all: build

build:
	$(MAKE) -C new_bloom_filter_repo_amd/csrc
	$(MAKE) -C oracle

test-cpu: build
	$(PY) -m pytest tests -q -m "not gpu"

test-gpu: build
	$(PY) -m pytest tests -q -m gpu

bench: build
	$(PY) bench.py

clean:
	$(MAKE) -C new_bloom_filter_repo_amd/csrc clean
	rm -rf build

.PHONY: all build test-cpu test-gpu bench clean
