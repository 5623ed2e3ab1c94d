/* A C99 caller of the boundary: links libgfdm_hip.so through include/gfdm_hip.h only (tests/test_boundary.py builds and runs it).
 * Without a GPU every create must fail with GFDM_HIP_ENODEV (no CPU fallback); with one, a K=4 M=3 block goes through the
 * modulator and the receiver (error codes, determinism of the arithmetic under scaling). */
#include <gfdm_hip.h>
#include <stdio.h>
#include <string.h>

int main(void)
{
    enum { M = 3, K = 4, L = 2, N = M * K };
    float taps[2 * L * M], in[2 * N], mid[2 * N], out[2 * N];
    int i, rc;
    gfdm_hip_modulator* mod = NULL;
    gfdm_hip_receiver* rx = NULL;
    gfdm_hip_resource_mapper* map = NULL;
    const int smap[2] = { 1, 3 };
    printf("version: %s\n", gfdm_hip_version());
    if (strcmp(gfdm_hip_strerror(0), "success") != 0) return 10;
    for (i = 0; i < L * M; ++i) { taps[2 * i] = (i < M) ? 1.0f : 0.25f; taps[2 * i + 1] = 0.0f; }
    for (i = 0; i < N; ++i) { in[2 * i] = (i % 3 == 0) ? 0.7071f : -0.7071f; in[2 * i + 1] = (i % 2) ? 0.7071f : -0.7071f; }
    rc = gfdm_hip_modulator_create(&mod, M, K, L, taps, L * M, 0);
    if (rc == GFDM_HIP_ENODEV) {
        if (mod != NULL) return 11;
        if (gfdm_hip_receiver_create(&rx, M, K, L, taps, L * M, 0) != GFDM_HIP_ENODEV) return 12;
        if (gfdm_hip_resource_mapper_create(&map, M, K, 2, smap, 2, 1, 0) != GFDM_HIP_ENODEV) return 13;
        printf("no device: %s\n", gfdm_hip_last_error());
        return 0;
    }
    if (rc != GFDM_HIP_OK) { printf("create: %s\n", gfdm_hip_last_error()); return 20; }
    if (gfdm_hip_receiver_create(&rx, M, K, L, taps, L * M, 0) != GFDM_HIP_OK) return 21;
    if (gfdm_hip_modulator_create(&mod, M, K, L, taps, L * M - 1, 0) != GFDM_HIP_EINVAL_TAPS) return 22;   /* wrong tap count */
    if (gfdm_hip_modulator_create(&mod, M, K, L, taps, L * M, 0) != GFDM_HIP_OK) return 23;
    if (gfdm_hip_modulator_work_host(mod, mid, in, 1) != GFDM_HIP_OK) return 24;
    if (gfdm_hip_receiver_demodulate_host(rx, out, mid, NULL, 1) != GFDM_HIP_OK) return 25;
    {   /* the receiver is linear: twice the samples, twice the symbols (exactly: a power of two) */
        float mid2[2 * N], out2[2 * N];
        float energy = 0.0f;
        for (i = 0; i < 2 * N; ++i) { mid2[i] = 2.0f * mid[i]; energy += out[i] * out[i]; }
        if (gfdm_hip_receiver_demodulate_host(rx, out2, mid2, NULL, 1) != GFDM_HIP_OK) return 26;
        for (i = 0; i < 2 * N; ++i)
            if (out2[i] != 2.0f * out[i]) { printf("not linear at %d: %g vs %g\n", i, out2[i], 2.0f * out[i]); return 27; }
        if (!(energy > 0.0f)) return 28;
    }
    gfdm_hip_modulator_destroy(mod);
    gfdm_hip_receiver_destroy(rx);
    printf("modulate + demodulate ok\n");
    return 0;
}
