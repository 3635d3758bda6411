/* The sharded bank over the C ABI, strict C99: one process, one handle over a list of devices (here the box's one GPU,
 * as one RCCL rank: `devices = {0}`; or listed `n` times -- "0,0,0" -- which rehearses the shard table, the time-axis split
 * and the copy exchange).  The reference is one process that owns every channel (Processor.swift:57-59,128-141;
 * main.swift:86-89,126-130); this program is what a host in C (or Swift through the bridging header) does with several GPUs.
 *
 * It runs the same recording through a plain bank (syldet_create on device 0) and through the sharded bank -- host buffers
 * (syldet_sharded_run) and device buffers with the one exchange (syldet_sharded_run_device) -- and demands identical bits.
 *
 * usage: sharded_bank net.txt samples.f32 channels devices      (samples: `channels` rows of raw fp32, equal lengths;
 *                                                                devices: comma-separated HIP device indices)
 * prints: "<E> <detections> <rccl ranks> identical"  |  "no-device"                                                     */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "syldet.h"

/* the HIP runtime's few entry points this program needs, declared here so that it stays a plain C translation unit */
extern int hipMalloc(void **p, size_t n);
extern int hipFree(void *p);
extern int hipMemcpy(void *dst, const void *src, size_t n, int kind);   /* 1 = host to device, 2 = device to host */
extern int hipSetDevice(int d);

#define CHECK(st) do { int s_ = (st); if (s_ != SYLDET_OK) { fprintf(stderr, "line %d: %s: %s\n", __LINE__, syldet_strerror(s_), syldet_last_error()); goto done; } } while (0)
#define HIP(e) do { int h_ = (e); if (h_ != 0) { fprintf(stderr, "line %d: HIP error %d\n", __LINE__, h_); goto done; } } while (0)

int main(int argc, char **argv)
{
    syldet_config_t *cfg = NULL;
    syldet_geometry_t geom;
    syldet_t *plain = NULL;
    syldet_sharded_t *bank = NULL;
    float *x = NULL, *want = NULL, *got = NULL;
    uint8_t *wfl = NULL, *gfl = NULL, *all = NULL;
    int32_t devices[64];
    void *d_x[64], *d_out[64], *d_fl[64], *d_all[64];
    int64_t strides[64];
    int n_dev = 0, C, i, st, rc = 1;
    int64_t S, E, total_det = 0;
    long bytes;
    char *tok;
    FILE *f;

    memset(d_x, 0, sizeof d_x); memset(d_out, 0, sizeof d_out); memset(d_fl, 0, sizeof d_fl); memset(d_all, 0, sizeof d_all);
    if (argc < 5) { fprintf(stderr, "usage: %s net.txt samples.f32 channels devices\n", argv[0]); return 2; }
    C = atoi(argv[3]);
    for (tok = strtok(argv[4], ","); tok && n_dev < 64; tok = strtok(NULL, ",")) devices[n_dev++] = atoi(tok);
    if (C < 1 || n_dev < 1) return 2;
    CHECK(syldet_config_load_text(argv[1], &cfg));
    CHECK(syldet_config_geometry(cfg, &geom));
    f = fopen(argv[2], "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", argv[2]); goto done; }
    fseek(f, 0, SEEK_END);
    bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    S = (int64_t)(bytes / 4) / C;
    x = (float *)malloc((size_t)C * (size_t)S * sizeof(float));
    if (!x || fread(x, sizeof(float), (size_t)C * (size_t)S, f) != (size_t)C * (size_t)S) { fclose(f); fprintf(stderr, "short read\n"); goto done; }
    fclose(f);

    st = syldet_create(cfg, C, devices[0], SYLDET_ENGINE_AUTO, &plain);
    if (st == SYLDET_ERR_NO_DEVICE) { printf("no-device\n"); rc = 0; goto done; }
    CHECK(st);
    E = syldet_count_evals(plain, S);
    if (E < 1) { fprintf(stderr, "recording too short\n"); goto done; }
    want = (float *)malloc((size_t)C * (size_t)E * (size_t)geom.outputs * sizeof(float));
    got = (float *)malloc((size_t)C * (size_t)E * (size_t)geom.outputs * sizeof(float));
    wfl = (uint8_t *)malloc((size_t)C * (size_t)E);
    gfl = (uint8_t *)malloc((size_t)C * (size_t)E);
    all = (uint8_t *)malloc((size_t)C * (size_t)E);
    if (!want || !got || !wfl || !gfl || !all) goto done;
    CHECK(syldet_run(plain, x, S, S, want, wfl));

    /* --- one handle over the listed devices --- */
    CHECK(syldet_create_sharded(cfg, C, devices, n_dev, SYLDET_ENGINE_AUTO, SYLDET_EXCHANGE_RCCL, &bank));
    if (syldet_sharded_channels(bank) != C || syldet_sharded_shards(bank) != n_dev) { fprintf(stderr, "bank shape\n"); goto done; }

    /* host buffers: the whole bank in one call, results straight into the caller's rows */
    memset(got, 0xff, (size_t)C * (size_t)E * (size_t)geom.outputs * sizeof(float));
    memset(gfl, 0xff, (size_t)C * (size_t)E);
    CHECK(syldet_sharded_run(bank, x, S, S, got, gfl));
    if (memcmp(got, want, (size_t)C * (size_t)E * (size_t)geom.outputs * sizeof(float)) != 0) { fprintf(stderr, "host call: outputs differ\n"); goto done; }
    if (memcmp(gfl, wfl, (size_t)C * (size_t)E) != 0) { fprintf(stderr, "host call: flags differ\n"); goto done; }

    /* device buffers: every shard's block on its device, the one exchange, every device receives every channel's flags */
    for (i = 0; i < n_dev; i++) {
        syldet_shard_t sh;
        int64_t s0, s1, e0, cnt, r;
        CHECK(syldet_sharded_shard(bank, i, &sh));
        CHECK(syldet_sharded_ranges(bank, i, S, &s0, &s1, &e0, &cnt));
        HIP(hipSetDevice(sh.device));
        strides[i] = s1 - s0;
        HIP(hipMalloc(&d_x[i], (size_t)sh.channels * (size_t)(s1 - s0) * sizeof(float) + 16));
        HIP(hipMalloc(&d_out[i], (size_t)sh.channels * (size_t)(cnt > 0 ? cnt : 1) * (size_t)geom.outputs * sizeof(float)));
        HIP(hipMalloc(&d_fl[i], (size_t)sh.channels * (size_t)(cnt > 0 ? cnt : 1)));
        HIP(hipMalloc(&d_all[i], (size_t)C * (size_t)E));
        for (r = 0; r < sh.channels; r++)
            HIP(hipMemcpy((float *)d_x[i] + r * (s1 - s0), x + (size_t)(sh.first_channel + r) * (size_t)S + s0, (size_t)(s1 - s0) * sizeof(float), 1));
    }
    CHECK(syldet_sharded_run_device(bank, (const float *const *)d_x, S, strides, (float *const *)d_out, (uint8_t *const *)d_fl, (uint8_t *const *)d_all));
    CHECK(syldet_sharded_synchronize(bank));
    for (i = 0; i < n_dev; i++) {
        syldet_shard_t sh;
        int64_t e0, cnt, r;
        CHECK(syldet_sharded_shard(bank, i, &sh));
        CHECK(syldet_sharded_ranges(bank, i, S, NULL, NULL, &e0, &cnt));
        HIP(hipSetDevice(sh.device));
        HIP(hipMemcpy(all, d_all[i], (size_t)C * (size_t)E, 2));
        if (memcmp(all, wfl, (size_t)C * (size_t)E) != 0) { fprintf(stderr, "device call: gathered flags on shard %d differ\n", i); goto done; }
        for (r = 0; r < sh.channels && cnt > 0; r++) {
            HIP(hipMemcpy(got, (float *)d_out[i] + r * cnt * geom.outputs, (size_t)cnt * (size_t)geom.outputs * sizeof(float), 2));
            if (memcmp(got, want + ((size_t)(sh.first_channel + r) * (size_t)E + (size_t)e0) * (size_t)geom.outputs, (size_t)cnt * (size_t)geom.outputs * sizeof(float)) != 0) {
                fprintf(stderr, "device call: outputs of shard %d row %d differ\n", i, (int)r);
                goto done;
            }
        }
    }
    for (i = 0; i < C * (int)E; i++) total_det += wfl[i];
    printf("%ld %ld %d identical\n", (long)E, (long)total_det, (int)syldet_sharded_rccl_ranks(bank));
    rc = 0;
done:
    for (i = 0; i < n_dev; i++) {
        if (d_x[i]) hipFree(d_x[i]);
        if (d_out[i]) hipFree(d_out[i]);
        if (d_fl[i]) hipFree(d_fl[i]);
        if (d_all[i]) hipFree(d_all[i]);
    }
    if (bank) syldet_sharded_destroy(bank);
    if (plain) syldet_destroy(plain);
    syldet_config_free(cfg);
    free(x); free(want); free(got); free(wfl); free(gfl); free(all);
    return rc;
}
