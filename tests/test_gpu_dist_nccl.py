"""RCCL (backend "nccl") inside the GPU test tier, at world size 1 (the GPU boxes have one device): the same
calls the N > 1 path makes -- OutboxGather (fixed-slot batched gather), gather_records (length-exchanging
container gather) and gather_device_records (exact-size point-to-point gather) -- run over RCCL on records the
GPU packed, and what rank 0 holds afterwards is parsed and compared with the coder's own rows.  Also runs
bench.py's two multi-GPU modes end to end with --force-dist."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %(repo)r)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", %(port)r)
import torch, torch.distributed as dist
torch.cuda.set_device(0)
device = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
dist.barrier()
from new_bloom_filter_repo_amd import _native as nat, dist as D
from new_bloom_filter_repo_amd.gop import GopCoder, torch_allocator
from new_bloom_filter_repo_amd.synthetic import make_gop
W, H, F = 320, 180, 6
n = W * H
stream = torch.cuda.Stream(device)
ctx = nat.Context(0, stream.cuda_stream)
coder = GopCoder(ctx, W, H, F, allocator=torch_allocator(device))
coder.load_frames(np.stack(make_gop(77, W, H, F, p=0.07)))
slot_words = (int(nat.lib().rbf_record_max_bytes(F - 1, n)) + 255) // 256 * 256 // 8
og = D.OutboxGather(slot_words, 2, device, streams=[stream])

class Slot:
    def __init__(self, t): self.ptr, self.nbytes = t.data_ptr(), t.numel() * 8
for s in range(5):                                 # 2 full outboxes + a partly filled one
    with torch.cuda.stream(stream):
        t = og.begin(0)
        coder.encode()
        coder.pack(Slot(t))
        og.end(0)
og.flush()
torch.cuda.synchronize()
rows = coder.results()
checked = 0
for ob in range(2):
    if og.sizes[ob] is None:
        continue
    for rec in og.received(ob, 0):
        assert rec.numel() == (D.record_used_bytes(rec[:32].cpu().numpy()) + 7) // 8 * 8
        for got, want in zip(D.unpack_device_record(rec, n), rows):
            assert got["l"] == want["l"] and got["witness_bits"] == want["witness_bits"]
            assert np.array_equal(got["witness"], want["witness"])
            if want["l"]:
                assert np.array_equal(got["filter"], want["filter"])
            checked += 1
# exact-size device gather
blk = coder.pack()
torch.cuda.synchronize()
got = D.gather_device_records([blk.tensor], device)
used = D.record_used_bytes(got[0][:32].cpu().numpy())
assert len(got) == 1 and got[0].numel() == (used + 7) // 8 * 8
for a, want in zip(D.unpack_device_record(got[0], n), rows):
    assert np.array_equal(a["witness"], want["witness"])
# container-record gather
recs = [(t, 2, bytes(rows[t]["witness"])) for t in range(F - 1)]
merged = D.gather_records(recs, device=device)
assert [(t, ty) for t, ty, _ in merged] == [(t, 2) for t in range(F - 1)] and all(b == recs[i][2] for i, (_, _, b) in enumerate(merged))
print(json.dumps({"slots_checked": checked, "sent": og.sent}))
dist.destroy_process_group()
'''


def run(cmd, timeout=600):
    """Runs the command, returns its JSON result line (RCCL itself prints to stdout, e.g. 'Librccl path : ...')."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=REPO)
    assert p.returncode == 0, p.stdout[-2000:] + "\n" + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert lines, p.stdout[-2000:]
    return json.loads(lines[-1])


def test_rccl_world1_gathers_records_the_gpu_packed():
    port = str(36000 + os.getpid() % 2000)
    res = run([sys.executable, "-c", WORKER % {"repo": REPO, "port": port}])
    assert res["sent"] == 3 and res["slots_checked"] == 3 * 5        # the last two exchanges: 2 + 1 records of 5 frames


def test_bench_force_dist_weak_mode():
    os.environ["MASTER_PORT"] = str(38000 + os.getpid() % 2000)
    res = run([sys.executable, "bench.py", "--force-dist", "--steps", "12", "--warmup", "2", "--width", "640", "--height", "360", "--frames", "8",
               "--gather-every", "4", "--no-cpu-baseline", "--exact-steps", "--clip-leg-frames", "40", "--keyframe-interval", "10", "--clip-steps", "2"], timeout=900)
    assert res["config"]["gather_to_rank0"] and res["verified_vs_oracle"]["frames"] == 7 * 4 * 3 and res["value"] > 0       # 4 pipelines x 3 resident GOPs
    assert res["config"]["gathered_records_parsed_on_rank0"] >= 4 and res["config"]["rccl_ranks"] == 1
    # the clip legs ride in the same line (BASELINE configs 3 and 5), with and without the gather
    for key in ("clip300", "clip300_uint16"):
        leg = res[key]
        assert leg["scaling"] == "strong" and leg["gather_to_rank0"] and leg["value"] > 0 and leg["without_gather"]["value"] > 0
        assert leg["verified_vs_oracle"]["frames"] == 36 == leg["verified_vs_oracle"]["of"] == leg["verified_vs_oracle"]["records_parsed_on_rank0"]
        assert leg["verified_vs_oracle"]["bytes_gathered_on_rank0"] > 0


def test_bench_own_launcher_world1():
    """`python bench.py --gpus N` starts its ranks itself: the same launch path at N = 1 (--spawn), RCCL initialised in the child."""
    res = run([sys.executable, "bench.py", "--spawn", "--force-dist", "--steps", "8", "--warmup", "2", "--width", "640", "--height", "360", "--frames", "6",
               "--gather-every", "4", "--no-cpu-baseline", "--exact-steps", "--no-clips"], timeout=900)
    assert res["config"]["launcher"] == "self-spawned" and res["config"]["rccl_ranks"] == 1 and res["n_gpus"] == 1
    assert res["verified_vs_oracle"]["frames"] == 5 * 4 * 3


def test_bench_clip_mode_strong_scaling_world1():
    os.environ["MASTER_PORT"] = str(40000 + os.getpid() % 2000)
    res = run([sys.executable, "bench.py", "--force-dist", "--clip-frames", "40", "--keyframe-interval", "10", "--steps", "2", "--warmup", "1",
               "--width", "640", "--height", "360"], timeout=900)
    assert res["scaling"] == "strong" and res["verified_vs_oracle"]["frames"] == 36 == res["verified_vs_oracle"]["of"]
    assert res["verified_vs_oracle"]["records_parsed_on_rank0"] == 36
    res = run([sys.executable, "bench.py", "--clip-frames", "21", "--keyframe-interval", "10", "--steps", "2", "--warmup", "1",
               "--width", "640", "--height", "360", "--bits", "16"], timeout=900)
    assert res["verified_vs_oracle"]["frames"] == 18


def test_bench_config3_at_size_world1():
    """BASELINE config 3 at its stated size (1920x1080, 300 frames, keyframe every 30) through RCCL at world 1: all 290 inter-frames
    against the oracle, 290 records parsed on rank 0; config 5's single-GPU half (16-bit) at 60 frames."""
    os.environ["MASTER_PORT"] = str(42000 + os.getpid() % 2000)
    res = run([sys.executable, "bench.py", "--force-dist", "--clip-frames", "300", "--keyframe-interval", "30", "--steps", "3", "--warmup", "1"], timeout=900)
    v = res["verified_vs_oracle"]
    assert v["frames"] == 290 == v["of"] == v["records_parsed_on_rank0"] and res["config"]["gather_to_rank0"]
    res = run([sys.executable, "bench.py", "--force-dist", "--clip-frames", "60", "--keyframe-interval", "30", "--steps", "3", "--warmup", "1", "--bits", "16"], timeout=900)
    v = res["verified_vs_oracle"]
    assert v["frames"] == 58 == v["of"] == v["records_parsed_on_rank0"]
