/* A C host on the C ABI of libseld_hip.so: no Python, no torch in the process.
 *
 *   gcc -std=c99 -O2 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/c_host_train_step.c \
 *       -L seld_amd -lseld_hip -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/seld_amd -o /tmp/c_host_train_step
 *   /tmp/c_host_train_step B T n_steps in.bin out.bin
 *
 * What a maintainer of the reference would write around train.trainstep (train.py:22-36) if the host were C: build model_config/seldnet.json's
 * seld_arch, seld_create, load weights (in.bin: [nparam weights | nstate BatchNorm moving statistics | x [B,T,64,7] | y_sed [B,S,12] |
 * y_doa [B,S,36]], fp32), run n_steps of seld_train_step with loss_weight 1,1000 and lr 1e-3, and write
 * out.bin: [sed [B,S,12] | doa [B,S,36] of the LAST step | sloss | dloss [B,S] | the updated weights | the gradient of the last step].
 * tests/test_model_gpu.py::test_c_host_drives_the_train_step compares out.bin with the fp64 oracle. */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "seld_hip.h"

#define CHECK_HIP(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); return 2; } } while (0)
#define CHECK_SELD(e) do { int r_ = (e); if (r_ != SELD_OK) { fprintf(stderr, "%s: %d %s\n", #e, r_, seld_last_error(ctx)); return 3; } } while (0)

static float* to_device(const float* host, size_t n) {
    float* d = NULL;
    if (hipMalloc((void**)&d, n * sizeof(float)) != hipSuccess) return NULL;
    if (host && hipMemcpy(d, host, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return NULL;
    return d;
}

int main(int argc, char** argv) {
    if (argc != 6) { fprintf(stderr, "usage: %s B T n_steps in.bin out.bin\n", argv[0]); return 1; }
    const int B = atoi(argv[1]), T = atoi(argv[2]), n_steps = atoi(argv[3]);
    seld_ctx* ctx = NULL;
    int32_t sizes[2] = {0, 0};
    if (seld_abi_sizes(sizes, 2) != 2 || sizes[0] != (int32_t)sizeof(seld_arch) || sizes[1] != (int32_t)sizeof(seld_loss_cfg)) {
        fprintf(stderr, "header and library disagree about the ABI structs\n");
        return 1;
    }
    /* model_config/seldnet.json; n_classes 12 (train.py:306-307) */
    seld_arch a;
    memset(&a, 0, sizeof a);
    a.in_ch = 7; a.n_freq = 64; a.n_conv = 3;
    for (int i = 0; i < 3; ++i) a.filters[i] = 64;
    a.pool_t[0] = 5; a.pool_f[0] = 4; a.pool_t[1] = 1; a.pool_f[1] = 4; a.pool_t[2] = 1; a.pool_f[2] = 2;
    a.n_gru = 2; a.gru_units[0] = a.gru_units[1] = 128;
    a.n_sed_dense = 1; a.sed_units[0] = 128; a.n_doa_dense = 1; a.doa_units[0] = 128;
    a.n_classes = 12;
    if (seld_create(&a, B, T, SELD_DTYPE_F32, 0, &ctx) != SELD_OK) { fprintf(stderr, "seld_create: %s\n", seld_last_error(NULL)); return 3; }
    const int S = T / 5, nc = 12;
    const size_t np = (size_t)seld_param_count(ctx), ns = (size_t)seld_state_count(ctx);
    const size_t nx = (size_t)B * T * 64 * 7, nys = (size_t)B * S * nc, nyd = 3 * nys, nrow = (size_t)B * S;
    const size_t n_in = np + ns + nx + nys + nyd;
    float* in = (float*)malloc(n_in * sizeof(float));
    FILE* f = fopen(argv[4], "rb");
    if (!in || !f || fread(in, sizeof(float), n_in, f) != n_in) { fprintf(stderr, "cannot read %zu floats from %s\n", n_in, argv[4]); return 1; }
    fclose(f);
    CHECK_SELD(seld_set_weights_host(ctx, in, (int64_t)np));
    CHECK_SELD(seld_set_state_host(ctx, in + np, (int64_t)ns));
    float *x = to_device(in + np + ns, nx), *ys = to_device(in + np + ns + nx, nys), *yd = to_device(in + np + ns + nx + nys, nyd);
    float *sed = to_device(NULL, nys), *doa = to_device(NULL, nyd), *sl = to_device(NULL, 1), *dl = to_device(NULL, nrow);
    if (!x || !ys || !yd || !sed || !doa || !sl || !dl) { fprintf(stderr, "hipMalloc / hipMemcpy failed\n"); return 2; }
    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));
    CHECK_SELD(seld_set_stream(ctx, st));
    seld_loss_cfg cfg = {SELD_DOA_MSE, 1.f, 1000.f, 1.f, 0.f};
    for (int i = 0; i < n_steps; ++i) CHECK_SELD(seld_train_step(ctx, x, ys, yd, &cfg, 1e-3f, 0, sed, doa, sl, dl));
    CHECK_SELD(seld_sync(ctx));
    const size_t n_out = nys + nyd + 1 + nrow + 2 * np;
    float* out = (float*)malloc(n_out * sizeof(float));
    if (!out) return 1;
    CHECK_HIP(hipMemcpy(out, sed, nys * sizeof(float), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(out + nys, doa, nyd * sizeof(float), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(out + nys + nyd, sl, sizeof(float), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(out + nys + nyd + 1, dl, nrow * sizeof(float), hipMemcpyDeviceToHost));
    CHECK_SELD(seld_get_weights_host(ctx, out + nys + nyd + 1 + nrow, (int64_t)np));
    CHECK_SELD(seld_get_grads_host(ctx, out + nys + nyd + 1 + nrow + np, (int64_t)np));
    f = fopen(argv[5], "wb");
    if (!f || fwrite(out, sizeof(float), n_out, f) != n_out) { fprintf(stderr, "cannot write %s\n", argv[5]); return 1; }
    fclose(f);
    printf("c host: %d step(s) of [%d, %d, 64, 7]: BCE %.6f, %zu parameters\n", n_steps, B, T, out[nys + nyd], np);
    seld_destroy(ctx);
    hipFree(x); hipFree(ys); hipFree(yd); hipFree(sed); hipFree(doa); hipFree(sl); hipFree(dl);
    free(in); free(out);
    return 0;
}
