/* A C consumer that drives the GPU path end to end through include/rbf.h: packed mask in (file), filter geometry
 * from rbf_plan_batch, insert + query/witness on the device, decode back, results out (file).
 * usage: abi_gpu <mask.bin> <n> <ones> <out.bin>      -- built and run by tests/test_gpu_integration_stub.py */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "rbf.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != RBF_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, rbf_last_error()); return 1; } } while (0)

int main(int argc, char **argv)
{
    if (argc != 5) return 2;
    const uint64_t n = strtoull(argv[2], NULL, 10), ones = strtoull(argv[3], NULL, 10);
    const uint64_t stride = (n + 63) / 64 * 8;
    uint8_t *mask = calloc(1, stride);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(mask, 1, (n + 7) / 8, f) != (n + 7) / 8) return 3;
    fclose(f);

    rbf_filter_params par;
    double k = 0;
    CHECK(rbf_plan_batch(n, &ones, 1, 1, &par, &k));
    if (par.m == 0) { fprintf(stderr, "passthrough frame\n"); return 4; }
    const rbf_seeds seeds = {0x12345678ull, 0x87654321ull, 999ull};          /* improved_video_compressor.py:62-63,94 */
    const uint64_t fstride = ((uint64_t)par.m + 63) / 64 * 8;

    rbf_ctx *ctx = NULL;
    CHECK(rbf_ctx_create(0, NULL, &ctx));
    void *d_mask, *d_filter, *d_wit, *d_stats, *d_back;
    CHECK(rbf_malloc(ctx, stride, &d_mask));
    CHECK(rbf_malloc(ctx, fstride, &d_filter));
    CHECK(rbf_malloc(ctx, stride, &d_wit));
    CHECK(rbf_malloc(ctx, 8 * RBF_STATS_PER_FRAME, &d_stats));
    CHECK(rbf_malloc(ctx, stride, &d_back));
    CHECK(rbf_memcpy_h2d(ctx, d_mask, mask, stride));
    CHECK(rbf_bloom_encode_batch(ctx, d_mask, stride, n, 1, &par, &seeds, d_filter, fstride, d_wit, stride, (uint64_t *)d_stats));
    CHECK(rbf_bloom_decode_batch(ctx, d_filter, fstride, d_wit, stride, n, 1, &par, &seeds, d_back, stride));
    uint64_t stats[RBF_STATS_PER_FRAME];
    uint8_t *filter = malloc(fstride), *wit = malloc(stride), *back = malloc(stride);
    CHECK(rbf_memcpy_d2h(ctx, stats, d_stats, sizeof stats));
    CHECK(rbf_memcpy_d2h(ctx, filter, d_filter, fstride));
    CHECK(rbf_memcpy_d2h(ctx, wit, d_wit, stride));
    CHECK(rbf_memcpy_d2h(ctx, back, d_back, stride));
    if (memcmp(back, mask, (n + 7) / 8) != 0) { fprintf(stderr, "decode(encode(mask)) != mask\n"); return 5; }

    /* out: m, floor_k, threshold, k bits, witness bits, filter ones, then filter bytes, then witness bytes */
    FILE *o = fopen(argv[4], "wb");
    uint64_t head[6] = {par.m, par.floor_k, par.threshold, 0, stats[RBF_STAT_WITNESS_BITS], stats[RBF_STAT_FILTER_ONES]};
    memcpy(&head[3], &k, 8);
    fwrite(head, 8, 6, o);
    fwrite(filter, 1, ((uint64_t)par.m + 7) / 8, o);
    fwrite(wit, 1, (stats[RBF_STAT_WITNESS_BITS] + 7) / 8, o);
    fclose(o);
    CHECK(rbf_free(ctx, d_mask)); CHECK(rbf_free(ctx, d_filter)); CHECK(rbf_free(ctx, d_wit)); CHECK(rbf_free(ctx, d_stats)); CHECK(rbf_free(ctx, d_back));
    CHECK(rbf_ctx_destroy(ctx));
    printf("ok m=%u floor_k=%u witness_bits=%" PRIu64 "\n", par.m, par.floor_k, stats[RBF_STAT_WITNESS_BITS]);
    return 0;
}
