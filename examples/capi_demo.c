/* Minimal C (not C++) caller of the drop-in boundary: one fused Theta + residual + loss + gradient pass over a
 * damped-oscillator batch through libsymode_hip.so.  Shows what a non-Python host (or another language's FFI) binds:
 * plain device pointers, sizes, a caller-provided workspace, int status codes.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/capi_demo.c \
 *       -Lsymmetry-ode-discovery_amd -lsymode_hip -L/opt/rocm/lib -lamdhip64 -o capi_demo                (needs a GPU to run)
 *   gcc -std=c99 -fsyntax-only -Iinclude -DSYMODE_DEMO_NO_HIP examples/capi_demo.c                       (header check, no ROCm)
 */
#include <stdio.h>
#include <stdlib.h>

#include "symode.h"

#ifndef SYMODE_DEMO_NO_HIP
#include <hip/hip_runtime_api.h>
#define CHECK_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)
#endif

int main(void) {
    const int d = 2, order = 3, flags = 0;
    const long n = 50L * 2500L;
    (void)n;
    const int p = symode_lib_size(d, order, flags);
    if (p < 0) {
        fprintf(stderr, "library not compiled in: %s\n", symode_error_string(p));
        return 1;
    }
    printf("ABI %d, library d=%d order=%d -> %d terms\n", symode_abi_version(), d, order, p);
#ifndef SYMODE_DEMO_NO_HIP
    float *hx = (float*)malloc(sizeof(float) * n * d), *hdx = (float*)malloc(sizeof(float) * n * d);
    float *hxi = (float*)calloc((size_t)d * p, sizeof(float));
    /* points on decaying spirals, exact derivative of dx0 = -0.1 x0 - x1, dx1 = x0 - 0.1 x1 */
    for (long i = 0; i < n; ++i) {
        const float a = 0.5f + (float)(i % 977) / 977.0f, b = -1.0f + 2.0f * (float)(i % 613) / 613.0f;
        hx[2 * i] = a; hx[2 * i + 1] = b;
        hdx[2 * i] = -0.1f * a - b; hdx[2 * i + 1] = a - 0.1f * b;
    }
    hxi[0 * p + 1] = -0.1f; hxi[0 * p + 2] = -1.0f; hxi[1 * p + 1] = 1.0f; hxi[1 * p + 2] = -0.05f;   /* one coefficient off by 0.05 */
    float *x, *dx, *xi, *loss, *grad;
    void* ws;
    const size_t ws_bytes = symode_workspace_bytes(d, order, flags, 1, n);
    CHECK_HIP(hipMalloc((void**)&x, sizeof(float) * n * d));
    CHECK_HIP(hipMalloc((void**)&dx, sizeof(float) * n * d));
    CHECK_HIP(hipMalloc((void**)&xi, sizeof(float) * d * p));
    CHECK_HIP(hipMalloc((void**)&loss, sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&grad, sizeof(float) * d * p));
    CHECK_HIP(hipMalloc(&ws, ws_bytes));
    if (symode_workspace_init(ws, ws_bytes, NULL) != SYMODE_OK) {      /* once per allocation */
        fprintf(stderr, "symode_workspace_init failed\n");
        return 1;
    }
    CHECK_HIP(hipMemcpy(x, hx, sizeof(float) * n * d, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dx, hdx, sizeof(float) * n * d, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(xi, hxi, sizeof(float) * d * p, hipMemcpyHostToDevice));
    const int rc = symode_loss_grad(x, dx, 1, n, d, order, flags, xi, NULL, 1.0f / (float)(n * d), loss, grad, ws, ws_bytes, NULL);
    if (rc != SYMODE_OK) {
        fprintf(stderr, "symode_loss_grad: %s\n", symode_error_string(rc));
        return 1;
    }
    float hloss, *hgrad = (float*)malloc(sizeof(float) * d * p);
    CHECK_HIP(hipMemcpy(&hloss, loss, sizeof(float), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(hgrad, grad, sizeof(float) * d * p, hipMemcpyDeviceToHost));
    printf("loss %.6e   dloss/dXi[1][2] %.6e (the perturbed coefficient)\n", hloss, hgrad[1 * p + 2]);
#endif
    return 0;
}
