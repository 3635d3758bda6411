/* include/syldet.h is a C header: this translation unit is compiled as strict C99 (-std=c99 -pedantic -Werror) and
 * linked against libsyldet, the way the reference's bridging header (Common/Common-Bridging-Header.h:5) pulls a C API
 * into the Swift project.  It goes through the whole life cycle of a detector bank with plain C types only:
 * load_text -> geometry -> create -> run -> detections -> destroy.  On a machine without a gfx950 device syldet_create
 * must fail with SYLDET_ERR_NO_DEVICE (there is no CPU fallback) and the program says so.
 *
 * usage: header_is_c net.txt samples.f32 [outputs.f32]      (one channel of raw little-endian fp32 samples) */
#include <stdio.h>
#include <stdlib.h>

#include "syldet.h"

int main(int argc, char **argv)
{
    syldet_config_t *cfg = NULL;
    syldet_geometry_t geom;
    syldet_t *h = NULL;
    float *x = NULL, *out = NULL;
    uint8_t *flags = NULL;
    int64_t *idx = NULL, count = 0, S, E;
    long bytes;
    FILE *f;
    int st, rc = 1;

    if (argc < 3) { fprintf(stderr, "usage: %s net.txt samples.f32 [outputs.f32]\n", argv[0]); return 2; }
    if (syldet_abi_version() != SYLDET_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 2; }
    st = syldet_config_load_text(argv[1], &cfg);
    if (st != SYLDET_OK) { fprintf(stderr, "%s: %s\n", syldet_strerror(st), syldet_last_error()); return 2; }
    st = syldet_config_geometry(cfg, &geom);
    if (st != SYLDET_OK) { fprintf(stderr, "%s: %s\n", syldet_strerror(st), syldet_last_error()); goto done; }
    f = fopen(argv[2], "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", argv[2]); goto done; }
    fseek(f, 0, SEEK_END);
    bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    S = (int64_t)(bytes / 4);
    x = (float *)malloc((size_t)(S > 0 ? S : 1) * sizeof(float));
    if (!x || fread(x, sizeof(float), (size_t)S, f) != (size_t)S) { fclose(f); fprintf(stderr, "short read\n"); goto done; }
    fclose(f);

    st = syldet_create(cfg, 1, 0, SYLDET_ENGINE_AUTO, &h);
    if (st == SYLDET_ERR_NO_DEVICE) { printf("no-device\n"); rc = 0; goto done; }
    if (st != SYLDET_OK) { fprintf(stderr, "%s: %s\n", syldet_strerror(st), syldet_last_error()); goto done; }
    E = syldet_count_evals(h, S);
    out = (float *)malloc((size_t)(E > 0 ? E : 1) * (size_t)geom.outputs * sizeof(float));
    flags = (uint8_t *)malloc((size_t)(E > 0 ? E : 1));
    idx = (int64_t *)malloc((size_t)(E > 0 ? E : 1) * sizeof(int64_t));
    if (!out || !flags || !idx) goto done;
    st = syldet_run(h, x, S, S, out, flags);
    if (st == SYLDET_OK && E > 0) st = syldet_detections(h, flags, E, 0.0, idx, E, &count);
    if (st != SYLDET_OK) { fprintf(stderr, "%s: %s\n", syldet_strerror(st), syldet_last_error()); goto done; }
    printf("%ld %ld %ld\n", (long)E, (long)count, (long)(count > 0 ? idx[0] : -1));
    if (argc > 3 && E > 0) {
        f = fopen(argv[3], "wb");
        if (!f || fwrite(out, sizeof(float), (size_t)E * (size_t)geom.outputs, f) != (size_t)E * (size_t)geom.outputs) { fprintf(stderr, "cannot write\n"); if (f) fclose(f); goto done; }
        fclose(f);
    }
    rc = 0;
done:
    if (h) syldet_destroy(h);
    syldet_config_free(cfg);
    free(x); free(out); free(flags); free(idx);
    return rc;
}
