// probe_styl16.hip - the packed-fp16 stylization chain of dc_dev.h (styl_tile under -DDC_STYL_PK16=1|2) against the fp32 form on random
// tiles, element by element (the chain inside k_layer gave wrong x0 on the box while tools/probe_pk16 - its SiLU part alone - was right).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDC_STYL_PK16=1 -I diffusion-conductor_amd/csrc -I include tools/probe_styl16.hip -o tools/probe_styl16_1
#include "dc_dev.h"
#include <cmath>
#include <cstdio>
#include <vector>
using namespace dc;

__global__ void k(const f16x16* y, const f16x16* g, const f16x16* h, const float* rstd, const float* shift, f16x8* z_pk, f16x8* z_ref, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    XFrag<_Float16, false> a, b;
    styl_tile<_Float16, false, f16x16, true>(a, y[i], rstd[i], shift[i], g[i], h[i]);        // packed-fp16 chain (G1 tiles)
    {   // the production form: three mixed-precision FMAs per element, SiLU in fp32
        f32x16 z;
        const u32x8 yw = __builtin_bit_cast(u32x8, y[i]), gw = __builtin_bit_cast(u32x8, g[i]), hw = __builtin_bit_cast(u32x8, h[i]);
        for (int k = 0; k < 8; ++k) {
            const float n0 = fma_mix_h<0>(yw[k], rstd[i], shift[i]), n1 = fma_mix_h<1>(yw[k], rstd[i], shift[i]);
            const f32x2 zz = silu_l2_pair(fma_mix_hh<0>(gw[k], n0, hw[k]), fma_mix_hh<1>(gw[k], n1, hw[k]));
            z[2 * k] = zz.x;
            z[2 * k + 1] = zz.y;
        }
        make_frag<_Float16, false>(z, b);
    }
    z_pk[2 * i] = a.hi[0];
    z_pk[2 * i + 1] = a.hi[1];
    z_ref[2 * i] = b.hi[0];
    z_ref[2 * i + 1] = b.hi[1];
}

int main() {
    const int n = 1 << 16;
    std::vector<_Float16> y(16 * n), g(16 * n), h(16 * n), zp(16 * n), zr(16 * n);
    std::vector<float> rs(n), sh(n);
    srand(3);
    auto rnd = []() { return (float)rand() / RAND_MAX - 0.5f; };
    for (int i = 0; i < n; ++i) {
        const float mean = 4.f * rnd(), sd = 0.2f + 3.f * fabsf(rnd());
        rs[i] = 1.4426950408889634f / sd;
        sh[i] = -mean * rs[i];
        for (int r = 0; r < 16; ++r) {
            y[16 * i + r] = (_Float16)(mean + sd * 4.f * rnd());
            g[16 * i + r] = (_Float16)(1.f + 1.5f * rnd());
            h[16 * i + r] = (_Float16)(3.f * rnd());
        }
    }
    f16x16 *dy, *dg, *dh;
    f16x8 *dzp, *dzr;
    float *drs, *dsh;
    (void)hipMalloc(&dy, 32 * n); (void)hipMalloc(&dg, 32 * n); (void)hipMalloc(&dh, 32 * n);
    (void)hipMalloc(&dzp, 32 * n); (void)hipMalloc(&dzr, 32 * n); (void)hipMalloc(&drs, 4 * n); (void)hipMalloc(&dsh, 4 * n);
    (void)hipMemcpy(dy, y.data(), 32 * n, hipMemcpyHostToDevice); (void)hipMemcpy(dg, g.data(), 32 * n, hipMemcpyHostToDevice);
    (void)hipMemcpy(dh, h.data(), 32 * n, hipMemcpyHostToDevice); (void)hipMemcpy(drs, rs.data(), 4 * n, hipMemcpyHostToDevice);
    (void)hipMemcpy(dsh, sh.data(), 4 * n, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dy, dg, dh, drs, dsh, dzp, dzr, n);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
    (void)hipMemcpy(zp.data(), dzp, 32 * n, hipMemcpyDeviceToHost); (void)hipMemcpy(zr.data(), dzr, 32 * n, hipMemcpyDeviceToHost);
    double num = 0, den = 0, worst = 0; long bad = 0; int shown = 0;
    for (int i = 0; i < 16 * n; ++i) {
        const double a = (float)zp[i], b = (float)zr[i];
        num += (a - b) * (a - b); den += b * b;
        const double e = fabs(a - b) / fmax(fabs(b), 0.05);
        if (e > worst) worst = e;
        if (!(e < 2e-2)) { ++bad; if (shown++ < 12) printf("  elem %d (tile %d, reg %d): packed %.5f  fp32 form %.5f\n", i, i / 16, i % 16, a, b); }
    }
    printf("DC_STYL_PK16=%d: rel-L2 packed vs fp32 form %.3e, worst element %.3e, elements off by more than 2e-2: %ld of %d\n", DC_STYL_PK16, sqrt(num / den), worst, bad, 16 * n);
    return bad ? 1 : 0;
}
