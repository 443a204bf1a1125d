/*
 * Torch-free use of the C ABI (include/hydro.h): plain C host code, HIP runtime for memory only.
 *
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ examples/c_abi_demo.c -I/opt/rocm/include -I include \
 *       -L silver2_isaacsim_amd/lib -lhydro -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/silver2_isaacsim_amd/lib -Wl,-rpath,/opt/rocm/lib -o c_abi_demo
 *   ./c_abi_demo bodies.bin          (bodies.bin: int64 n, then n x 13 state, n x 6 prev, n x 11 params floats, float dt)
 *
 * Prints per-body wrench rows "Fx Fy Fz Tx Ty Tz" so that a test can compare them with the oracle.
 * It goes through the plain-SoA entry, then the tiled entry (repacked on device) and checks that the
 * two agree bit for bit, exercises the error paths (status codes + hydro_last_error), and runs a closed loop
 * of 64 fused steps from a HIP graph captured on its own stream (same bits as the eager loop).
 */
#include <hip/hip_runtime_api.h>
#ifdef HYDRO_DEMO_WITH_RCCL            /* add -DHYDRO_DEMO_WITH_RCCL -lrccl: the global kinetic energy from C */
#include <rccl/rccl.h>
#endif
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hydro.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_HYDRO(h, x) do { int rc_ = (x); if (rc_ != HYDRO_OK) { fprintf(stderr, "%s -> %s: %s\n", #x, hydro_status_string(rc_), hydro_last_error(h)); return 3; } } while (0)

static float *to_soa(const float *aos, int64_t n, int fields)
{
    float *soa = (float *)malloc(sizeof(float) * n * fields);
    for (int64_t i = 0; i < n; ++i)
        for (int f = 0; f < fields; ++f) soa[(int64_t)f * n + i] = aos[i * fields + f];
    return soa;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s bodies.bin\n", argv[0]); return 1; }
    FILE *fp = fopen(argv[1], "rb");
    if (!fp) { perror("open"); return 1; }
    int64_t n = 0;
    if (fread(&n, sizeof n, 1, fp) != 1 || n <= 0) return 1;
    float *state = malloc(sizeof(float) * n * 13), *prev = malloc(sizeof(float) * n * 6), *params = malloc(sizeof(float) * n * 11), dt = 0;
    if (fread(state, sizeof(float), n * 13, fp) != (size_t)(n * 13) || fread(prev, sizeof(float), n * 6, fp) != (size_t)(n * 6) ||
        fread(params, sizeof(float), n * 11, fp) != (size_t)(n * 11) || fread(&dt, sizeof dt, 1, fp) != 1) return 1;
    fclose(fp);

    fprintf(stderr, "libhydro version 0x%06x\n", hydro_version());
    hydro_t *h = NULL;
    /* error model first: status codes, never a crash */
    if (hydro_create(0, -1, &h) != HYDRO_E_ARG || hydro_create(999, 16, &h) != HYDRO_E_DEVICE) { fprintf(stderr, "error model broken\n"); return 4; }
    CHECK_HYDRO(h, hydro_create(0, n, &h));
    CHECK_HYDRO(h, hydro_set_scene(h, 1025.0, 9.81));

    /* device buffers: plain SoA, one run of n floats per field */
    float *s_soa = to_soa(state, n, 13), *p_soa = to_soa(prev, n, 6), *q_soa = to_soa(params, n, 11);
    float *d_state, *d_prev, *d_out, *d_out2;
    CHECK_HIP(hipMalloc((void **)&d_state, sizeof(float) * n * 13));
    CHECK_HIP(hipMalloc((void **)&d_prev, sizeof(float) * n * 6));
    CHECK_HIP(hipMalloc((void **)&d_out, sizeof(float) * n * 6));
    CHECK_HIP(hipMalloc((void **)&d_out2, sizeof(float) * n * 6));
    CHECK_HIP(hipMemcpy(d_state, s_soa, sizeof(float) * n * 13, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_prev, p_soa, sizeof(float) * n * 6, hipMemcpyHostToDevice));
    const float *st[HYDRO_STATE_FIELDS], *pv[HYDRO_PREV_FIELDS], *prm[HYDRO_PARAM_FIELDS];
    float *out[HYDRO_WRENCH_FIELDS], *out2[HYDRO_WRENCH_FIELDS];
    for (int f = 0; f < 13; ++f) st[f] = d_state + (int64_t)f * n;
    for (int f = 0; f < 6; ++f) { pv[f] = d_prev + (int64_t)f * n; out[f] = d_out + (int64_t)f * n; out2[f] = d_out2 + (int64_t)f * n; }
    for (int f = 0; f < 11; ++f) prm[f] = q_soa + (int64_t)f * n;                 /* host arrays: on_device = 0 */

    /* a step before the parameters are set is a state error, with a message */
    if (hydro_step_wrench_ext(h, n, st, pv, dt, out, NULL) != HYDRO_E_STATE) { fprintf(stderr, "expected HYDRO_E_STATE\n"); return 4; }
    fprintf(stderr, "expected failure reported as: %s\n", hydro_last_error(h));
    CHECK_HYDRO(h, hydro_set_params_f32(h, n, prm, 0));

    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    CHECK_HYDRO(h, hydro_step_wrench_ext(h, n, st, pv, dt, out, stream));

    /* the same step through the tiled (native) layout: repack on device, step, unpack */
    const int64_t tiles = (n + HYDRO_TILE - 1) / HYDRO_TILE;
    float *t_state, *t_prev, *t_out;
    CHECK_HIP(hipMalloc((void **)&t_state, sizeof(float) * tiles * 13 * HYDRO_TILE));
    CHECK_HIP(hipMalloc((void **)&t_prev, sizeof(float) * tiles * 6 * HYDRO_TILE));
    CHECK_HIP(hipMalloc((void **)&t_out, sizeof(float) * tiles * 6 * HYDRO_TILE));
    CHECK_HYDRO(h, hydro_repack(h, n, 13, (float *const *)st, t_state, 13 * HYDRO_TILE, 1, stream));
    CHECK_HYDRO(h, hydro_repack(h, n, 6, (float *const *)pv, t_prev, 6 * HYDRO_TILE, 1, stream));
    CHECK_HYDRO(h, hydro_step_wrench_tiled(h, n, t_state, 13 * HYDRO_TILE, t_prev, 6 * HYDRO_TILE, dt, t_out, 6 * HYDRO_TILE, stream));
    CHECK_HYDRO(h, hydro_repack(h, n, 6, out2, t_out, 6 * HYDRO_TILE, 0, stream));
    CHECK_HIP(hipStreamSynchronize(stream));

    float *w = malloc(sizeof(float) * n * 6), *w2 = malloc(sizeof(float) * n * 6);
    CHECK_HIP(hipMemcpy(w, d_out, sizeof(float) * n * 6, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(w2, d_out2, sizeof(float) * n * 6, hipMemcpyDeviceToHost));
    if (memcmp(w, w2, sizeof(float) * n * 6) != 0) { fprintf(stderr, "tiled and plain-SoA entries disagree\n"); return 5; }

    /* several independent scenes in ONE launch (hydro_step_wrench_tiled_batch): here the same scene twice, into two
     * wrench buffers - each must carry the bits of the single launch above */
    {
        float *t_out_a, *t_out_b;
        const size_t wbytes = sizeof(float) * tiles * 6 * HYDRO_TILE;
        CHECK_HIP(hipMalloc((void **)&t_out_a, wbytes)); CHECK_HIP(hipMalloc((void **)&t_out_b, wbytes));
        hydro_scene_t scenes[2];
        for (int k = 0; k < 2; ++k) {
            scenes[k].engine = h; scenes[k].n = n;
            scenes[k].state = t_state; scenes[k].state_tile_stride = 13 * HYDRO_TILE;
            scenes[k].prev = t_prev; scenes[k].prev_tile_stride = 6 * HYDRO_TILE;
            scenes[k].wrench = k ? t_out_b : t_out_a; scenes[k].wrench_tile_stride = 6 * HYDRO_TILE;
        }
        CHECK_HYDRO(h, hydro_step_wrench_tiled_batch(2, scenes, dt, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        float *ref = malloc(wbytes), *ga = malloc(wbytes), *gb = malloc(wbytes);
        CHECK_HIP(hipMemcpy(ref, t_out, wbytes, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(ga, t_out_a, wbytes, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(gb, t_out_b, wbytes, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < n; ++i)
            for (int f = 0; f < 6; ++f) {
                const size_t at = (size_t)(i / HYDRO_TILE) * 6 * HYDRO_TILE + (size_t)f * HYDRO_TILE + (size_t)(i % HYDRO_TILE);
                if (memcmp(&ref[at], &ga[at], sizeof(float)) != 0 || memcmp(&ref[at], &gb[at], sizeof(float)) != 0) {
                    fprintf(stderr, "batched launch and single launch disagree at body %lld field %d\n", (long long)i, f); return 9;
                }
            }
        if (hydro_step_wrench_tiled_batch(0, scenes, dt, stream) != HYDRO_E_ARG || hydro_step_wrench_tiled_batch(HYDRO_BATCH_MAX + 1, scenes, dt, stream) != HYDRO_E_ARG) return 9;
        fprintf(stderr, "batched launch: 2 scenes in one launch, bit-identical to the single launch\n");
        free(ref); free(ga); free(gb);
        CHECK_HIP(hipFree(t_out_a)); CHECK_HIP(hipFree(t_out_b));
    }

    double ke[2];
    double *d_ke;
    CHECK_HIP(hipMalloc((void **)&d_ke, sizeof ke));
    CHECK_HYDRO(h, hydro_kinetic_energy(h, n, st, 1, d_ke, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    CHECK_HIP(hipMemcpy(ke, d_ke, sizeof ke, hipMemcpyDeviceToHost));
    fprintf(stderr, "kinetic energy: %.9e + %.9e J\n", ke[0], ke[1]);
#ifdef HYDRO_DEMO_WITH_RCCL
    /* the one collective of the path, from C: a communicator of this process's ranks (one here; one per GPU in a real
     * job), the pair summed in place on the stream.  With one rank the sum is the value itself. */
    {
        ncclComm_t comm;
        const int dev0 = 0;
        if (ncclCommInitAll(&comm, 1, &dev0) != ncclSuccess) { fprintf(stderr, "ncclCommInitAll failed\n"); return 8; }
        /* a communicator belongs to the copy of RCCL that made it: hand libhydro THIS program's ncclAllReduce */
        CHECK_HYDRO(h, hydro_bind_rccl((void *)ncclAllReduce, (void *)ncclGetErrorString));
        if (strcmp(hydro_rccl_origin(), "hydro_bind_rccl") != 0) return 8;
        CHECK_HYDRO(h, hydro_ke_allreduce(h, comm, d_ke, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        double all[2];
        CHECK_HIP(hipMemcpy(all, d_ke, sizeof all, hipMemcpyDeviceToHost));
        if (all[0] != ke[0] || all[1] != ke[1]) { fprintf(stderr, "all-reduce over one rank changed the value\n"); return 8; }
        if (hydro_ke_allreduce(h, NULL, d_ke, stream) != HYDRO_E_ARG) return 8;
        fprintf(stderr, "global kinetic energy over %d rank(s) through RCCL: %.9e + %.9e J\n", 1, all[0], all[1]);
        ncclCommDestroy(comm);
    }
#endif

    /* closed loop without a host round trip: 64 fused steps (wrench + integrator, ping-pong state buffers)
     * captured ONCE into a HIP graph and replayed - the step functions neither allocate nor synchronise, so
     * they are capture-safe.  The eager loop over the same 64 steps must give the same bits. */
    {
        enum { K = 64 };
        float *a0, *a1, *b0, *b1;
        const size_t sbytes = sizeof(float) * tiles * 13 * HYDRO_TILE;
        CHECK_HIP(hipMalloc((void **)&a0, sbytes)); CHECK_HIP(hipMalloc((void **)&a1, sbytes));
        CHECK_HIP(hipMalloc((void **)&b0, sbytes)); CHECK_HIP(hipMalloc((void **)&b1, sbytes));
        CHECK_HIP(hipMemcpyAsync(a0, t_state, sbytes, hipMemcpyDeviceToDevice, stream));
        CHECK_HIP(hipMemsetAsync(a1, 0, sbytes, stream));                       /* "previous" state: zero velocity */
        CHECK_HIP(hipMemcpyAsync(b0, a0, sbytes, hipMemcpyDeviceToDevice, stream));
        CHECK_HIP(hipMemcpyAsync(b1, a1, sbytes, hipMemcpyDeviceToDevice, stream));
        const int64_t ss = 13 * HYDRO_TILE;
        for (int k = 0; k < K; ++k) {                                           /* eager reference */
            float *cur = (k & 1) ? a1 : a0, *old = (k & 1) ? a0 : a1;
            CHECK_HYDRO(h, hydro_step_fused_tiled(h, n, cur, ss, old + 7 * HYDRO_TILE, ss, dt, old, ss, NULL, 0, 1, stream));
        }
        hipGraph_t graph; hipGraphExec_t exec;
        CHECK_HIP(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < K; ++k) {
            float *cur = (k & 1) ? b1 : b0, *old = (k & 1) ? b0 : b1;
            CHECK_HYDRO(h, hydro_step_fused_tiled(h, n, cur, ss, old + 7 * HYDRO_TILE, ss, dt, old, ss, NULL, 0, 1, stream));
        }
        CHECK_HIP(hipStreamEndCapture(stream, &graph));
        CHECK_HIP(hipGraphInstantiate(&exec, graph, NULL, NULL, 0));
        CHECK_HIP(hipGraphLaunch(exec, stream));                                /* one host call = 64 physics steps */
        CHECK_HIP(hipStreamSynchronize(stream));
        float *ha = malloc(sbytes), *hb = malloc(sbytes);
        CHECK_HIP(hipMemcpy(ha, a0, sbytes, hipMemcpyDeviceToHost));           /* K even: the newest state is in buffer 0 */
        CHECK_HIP(hipMemcpy(hb, b0, sbytes, hipMemcpyDeviceToHost));
        int same = 1, finite = 1;
        for (int64_t i = 0; i < n; ++i)
            for (int f = 0; f < 13; ++f) {
                const size_t at = (size_t)(i / HYDRO_TILE) * 13 * HYDRO_TILE + (size_t)f * HYDRO_TILE + (size_t)(i % HYDRO_TILE);
                if (memcmp(&ha[at], &hb[at], sizeof(float)) != 0) same = 0;
                if (!(hb[at] == hb[at]) || hb[at] > 1e30f || hb[at] < -1e30f) finite = 0;
            }
        if (!same || !finite) { fprintf(stderr, "graph replay and eager loop disagree (same=%d finite=%d)\n", same, finite); return 6; }
        /* ... and the same 64 steps as ONE launch: the bodies are independent, hydro_step_fused_tiled_multi carries each of
         * them through the steps in registers.  prev_out = the velocity fields of the state being read: after the call
         * (c1, c0) are the (current, previous) pair, exactly as after one single step. */
        {
            float *c0, *c1;
            CHECK_HIP(hipMalloc((void **)&c0, sbytes)); CHECK_HIP(hipMalloc((void **)&c1, sbytes));
            CHECK_HIP(hipMemcpy(c0, t_state, sbytes, hipMemcpyDeviceToDevice));
            CHECK_HIP(hipMemset(c1, 0, sbytes));
            CHECK_HYDRO(h, hydro_step_fused_tiled_multi(h, n, c0, ss, c1 + 7 * HYDRO_TILE, ss, dt, K - 1, c1, ss, c0 + 7 * HYDRO_TILE, ss, 1, 1, NULL, stream));
            CHECK_HYDRO(h, hydro_step_fused_tiled_multi(h, n, c1, ss, c0 + 7 * HYDRO_TILE, ss, dt, 1, c0, ss, c1 + 7 * HYDRO_TILE, ss, 1, 1, NULL, stream));
            CHECK_HIP(hipStreamSynchronize(stream));
            CHECK_HIP(hipMemcpy(hb, c0, sbytes, hipMemcpyDeviceToHost));       /* 63 + 1 steps: the newest state is in c0 */
            for (int64_t i = 0; i < n; ++i)
                for (int f = 0; f < 13; ++f) {
                    const size_t at = (size_t)(i / HYDRO_TILE) * 13 * HYDRO_TILE + (size_t)f * HYDRO_TILE + (size_t)(i % HYDRO_TILE);
                    if (memcmp(&ha[at], &hb[at], sizeof(float)) != 0) same = 0;
                }
            if (!same) { fprintf(stderr, "resident multi-step launch and eager loop disagree\n"); return 7; }
            if (hydro_step_fused_tiled_multi(h, n, c0, ss, c1 + 7 * HYDRO_TILE, ss, dt, 0, c1, ss, c0 + 7 * HYDRO_TILE, ss, 1, 1, NULL, stream) != HYDRO_E_ARG) return 7;
            fprintf(stderr, "resident loop: 63 + 1 steps in two launches, bit-identical to the eager loop\n");
            CHECK_HIP(hipFree(c0)); CHECK_HIP(hipFree(c1));
        }
        hipEvent_t e0, e1; float ms = 0.0f;
        CHECK_HIP(hipEventCreate(&e0)); CHECK_HIP(hipEventCreate(&e1));
        CHECK_HIP(hipEventRecord(e0, stream));
        for (int r = 0; r < 16; ++r) CHECK_HIP(hipGraphLaunch(exec, stream));
        CHECK_HIP(hipEventRecord(e1, stream));
        CHECK_HIP(hipEventSynchronize(e1));
        CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
        fprintf(stderr, "hip graph: %d fused steps per replay, bit-identical to the eager loop, %.2f us per physics step\n", K, ms * 1e3f / (16 * K));
        CHECK_HIP(hipGraphExecDestroy(exec)); CHECK_HIP(hipGraphDestroy(graph));
        free(ha); free(hb);
    }

    for (int64_t i = 0; i < n; ++i)
        printf("%.9e %.9e %.9e %.9e %.9e %.9e\n", w[0 * n + i], w[1 * n + i], w[2 * n + i], w[3 * n + i], w[4 * n + i], w[5 * n + i]);
    CHECK_HYDRO(h, hydro_destroy(h));
    return 0;
}
