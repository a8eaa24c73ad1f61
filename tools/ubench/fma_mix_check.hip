#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(float *out, float c, float acc) {
    unsigned b = threadIdx.x;           // 0..255
    unsigned packed = b | ((255u - b) << 16);   // two zero-extended bytes = two f16 denormals
    float lo, hi;
    float cs = c * 16777216.0f;
    asm volatile("v_fma_mix_f32 %0, %2, %3, %4 op_sel_hi:[1,0,0]\n"
                 "v_fma_mix_f32 %1, %2, %3, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                 : "=&v"(lo), "=&v"(hi) : "v"(packed), "v"(cs), "v"(acc));
    out[b] = lo; out[256 + b] = hi;
}
int main() {
    float *d; hipMalloc(&d, 512 * 4);
    float c = 0.299f, acc = 1.25f;
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, c, acc);
    float h[512]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < 256; b++) {
        float w0 = fmaf((float)b, c, acc), w1 = fmaf((float)(255 - b), c, acc);
        if (h[b] != w0 || h[256 + b] != w1) { if (bad < 5) printf("b=%d got %g %g want %g %g\n", b, h[b], h[256+b], w0, w1); bad++; }
    }
    printf("fma_mix denormal-half check: %d mismatches\n", bad);
    return 0;
}
