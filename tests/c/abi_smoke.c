/* A C consumer of include/rbf.h: host-side entry points only (no GPU needed).  Prints one line per query so the
 * Python test can compare with the fixtures.  Built and run by tests/test_host_cpu.py::test_c_consumer_links_and_agrees. */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include "rbf.h"

int main(int argc, char **argv)
{
    if (rbf_version() != RBF_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 2; }
    printf("version %d\n", rbf_version());
    for (int i = 1; i + 1 < argc; i += 2) {
        const uint64_t n = strtoull(argv[i], NULL, 10), ones = strtoull(argv[i + 1], NULL, 10);
        double k = 0;
        uint64_t l = 0, thr = 0;
        uint32_t fk = 0;
        if (rbf_optimal_params(n, ones, &k, &l) != RBF_OK) { fprintf(stderr, "%s\n", rbf_last_error()); return 3; }
        if (k > 0 && rbf_activation_threshold(k, &fk, &thr) != RBF_OK) { fprintf(stderr, "%s\n", rbf_last_error()); return 4; }
        rbf_filter_params p;
        double kk = 0;
        if (rbf_plan_batch(n, &ones, 1, 1, &p, &kk) != RBF_OK) { fprintf(stderr, "%s\n", rbf_last_error()); return 5; }
        printf("params %" PRIu64 " %" PRIu64 " %a %" PRIu64 " %u %" PRIu64 " plan %u %u %" PRIu64 "\n", n, ones, k, l, fk, thr, p.m, p.floor_k, p.threshold);
    }
    printf("record_max %" PRIu64 "\n", rbf_record_max_bytes(29, 2073600));
    if (rbf_optimal_params(0, 0, NULL, NULL) == RBF_OK) return 6;          /* errors come back as codes + a message */
    printf("error %s\n", rbf_last_error());
    return 0;
}
