// Does v_cvt_pk_f16_f32 (what hipcc emits for a vector float -> _Float16 conversion on gfx950) round like v_cvt_f16_f32 (scalar conversion)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(2))) _Float16 h2;
__global__ void k(const float *x, uint16_t *pk, uint16_t *sc, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    h2 v; v[0] = (_Float16)x[2 * i]; v[1] = (_Float16)x[2 * i + 1];              // packed form
    *reinterpret_cast<h2 *>(pk + 2 * i) = v;
    _Float16 a, b;
    asm volatile("v_cvt_f16_f32_e32 %0, %1" : "=v"(a) : "v"(x[2 * i]));
    asm volatile("v_cvt_f16_f32_e32 %0, %1" : "=v"(b) : "v"(x[2 * i + 1]));
    sc[2 * i] = __builtin_bit_cast(uint16_t, a); sc[2 * i + 1] = __builtin_bit_cast(uint16_t, b);
}
int main()
{
    const int n = 1 << 20;
    std::vector<float> h(n);
    uint32_t s = 12345;
    for (int i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; uint32_t bits = (s & 0x807FFFFFu) | ((100u + (s >> 8) % 40u) << 23); memcpy(&h[i], &bits, 4); }   // magnitudes 2^-27 .. 2^12: normals and fp16 denormals
    float *dx; uint16_t *dp, *ds;
    hipMalloc(&dx, n * 4); hipMalloc(&dp, n * 2); hipMalloc(&ds, n * 2);
    hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, dx, dp, ds, n);
    std::vector<uint16_t> p(n), q(n);
    hipMemcpy(p.data(), dp, n * 2, hipMemcpyDeviceToHost); hipMemcpy(q.data(), ds, n * 2, hipMemcpyDeviceToHost);
    int diff = 0, shown = 0;
    for (int i = 0; i < n; i++) if (p[i] != q[i]) { diff++; if (shown++ < 8) printf("x = %.9g  packed %04x  scalar %04x\n", h[i], p[i], q[i]); }
    printf("%d of %d conversions differ between v_cvt_pk_f16_f32 and v_cvt_f16_f32\n", diff, n);
    return 0;
}
