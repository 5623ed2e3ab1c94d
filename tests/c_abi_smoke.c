/* A C99 caller of the boundary: links libgfdm_hip.so through include/gfdm_hip.h only (tests/test_boundary.py builds and runs it).
 * Without a GPU every create must fail with GFDM_HIP_ENODEV (no CPU fallback); with one, a K=4 M=3 block goes through the
 * modulator and the receiver (error codes, determinism of the arithmetic under scaling), and a run of blocks goes through buffers
 * registered once with gfdm_hip_register_host -- what a GNU Radio block does with its long-lived circular buffers -- which the host
 * calls then use in place (gfdm_hip_host_call_stats), with the results of the bounced call. */
#define _POSIX_C_SOURCE 200112L      /* posix_memalign: page-aligned buffers for gfdm_hip_register_host */
#include <gfdm_hip.h>
#include <stdio.h>
#include <stdlib.h>
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
    {   /* a scheduler's buffers: allocated once, registered once, any run of blocks inside them per call */
        enum { NB = 700 };
        const size_t bytes = (size_t)NB * N * 2 * sizeof(float), pages = (bytes + 4095) / 4096 * 4096;    /* registration takes whole pages */
        void *ring_in_v = NULL, *ring_out_v = NULL;
        float *ring_in, *ring_out;
        float* plain_out = (float*)malloc(bytes);
        int64_t chunks = 0, staged = 0;
        unsigned direct = 0;
        if (posix_memalign(&ring_in_v, 4096, pages) != 0 || posix_memalign(&ring_out_v, 4096, pages) != 0 || !plain_out) return 30;
        ring_in = (float*)ring_in_v;
        ring_out = (float*)ring_out_v;
        for (i = 0; i < NB * N * 2; ++i) ring_in[i] = mid[i % (2 * N)] * (float)(1 + i % 5);
        if (gfdm_hip_receiver_demodulate_host(rx, plain_out, ring_in, NULL, NB) != GFDM_HIP_OK) return 31;          /* pageable: bounced */
        if (gfdm_hip_host_call_stats(&chunks, NULL, &staged, &direct, NULL, NULL) != GFDM_HIP_OK || direct != 0 || staged != (int64_t)(2 * bytes)) return 32;
        if (gfdm_hip_register_host(plain_out + 1, 4096) != GFDM_HIP_EINVAL) return 39;                             /* not whole pages: refused */
        if (gfdm_hip_register_host(ring_in, pages) != GFDM_HIP_OK || gfdm_hip_register_host(ring_out, pages) != GFDM_HIP_OK) { printf("register: %s\n", gfdm_hip_last_error()); return 33; }
        if (gfdm_hip_receiver_demodulate_host(rx, ring_out + 2 * N * 3, ring_in + 2 * N * 3, NULL, NB - 3) != GFDM_HIP_OK) return 34;   /* a run inside the buffers */
        if (gfdm_hip_host_call_stats(&chunks, NULL, &staged, &direct, NULL, NULL) != GFDM_HIP_OK || direct != 3u || staged != 0 || chunks != 1) return 35;
        if (memcmp(ring_out + 2 * N * 3, plain_out + 2 * N * 3, bytes - (size_t)3 * N * 2 * sizeof(float)) != 0) return 36;
        if (gfdm_hip_unregister_host(ring_in) != GFDM_HIP_OK || gfdm_hip_unregister_host(ring_out) != GFDM_HIP_OK) return 37;
        if (gfdm_hip_unregister_host(ring_in) == GFDM_HIP_OK) return 38;                                             /* not registered any more */
        free(ring_in); free(ring_out); free(plain_out);
    }
    gfdm_hip_modulator_destroy(mod);
    gfdm_hip_receiver_destroy(rx);
    printf("modulate + demodulate ok, registered buffers used in place\n");
    return 0;
}
