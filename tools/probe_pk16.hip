// probe_pk16.hip - do 16-bit transcendentals in SDWA form with dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE (read the high half, write the
// high half, keep the low half) behave on gfx950 as the packed-fp16 stylization chain of dc_dev.h (styl_tile, DC_STYL_PK16) needs?  hipcc
// itself extracts the high half with SDWA and re-packs with v_pack_b32_f16 (3 instructions per pair of transcendentals instead of 2;
// the VOP3 spelling with op_sel:[1,1] is rejected by the gfx950 assembler for VOP1 operations).
// Every lane runs SiLU on a pair of fp16 values in the three spellings and compares with the fp32 computation.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe_pk16.hip -o tools/probe_pk16
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

#define DEV __device__ __forceinline__
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
// back to back in one asm statement (what round 6's first build did)
DEV unsigned exp2neg_b2b(unsigned u) {
    unsigned e;
    asm("v_exp_f16_sdwa %0, -%1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0\n\t"
        "v_exp_f16_sdwa %0, -%1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "=&v"(e) : "v"(u));
    return e;
}
DEV unsigned rcp_b2b(unsigned u) {
    unsigned e;
    asm("v_rcp_f16_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0\n\t"
        "v_rcp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "=&v"(e) : "v"(u));
    return e;
}
// the same with one wait state between the two (s_nop 0)
DEV unsigned exp2neg_nop(unsigned u) {
    unsigned e;
    asm("v_exp_f16_sdwa %0, -%1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0\n\ts_nop 1\n\t"
        "v_exp_f16_sdwa %0, -%1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "=&v"(e) : "v"(u));
    return e;
}
DEV unsigned rcp_nop(unsigned u) {
    unsigned e;
    asm("v_rcp_f16_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0\n\ts_nop 1\n\t"
        "v_rcp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "=&v"(e) : "v"(u));
    return e;
}
// dc_dev.h's form: eight low halves, a scheduling barrier, eight high halves
#include <stdint.h>
DEV uint32_t exp2neg_lo16(uint32_t u) { uint32_t e; asm("v_exp_f16_sdwa %0, -%1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(e) : "v"(u)); return e; }
DEV void exp2neg_hi16(uint32_t& e, uint32_t u) { asm("v_exp_f16_sdwa %0, -%1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(e) : "v"(u)); }
DEV uint32_t rcp_lo16(uint32_t d) { uint32_t r; asm("v_rcp_f16_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(r) : "v"(d)); return r; }
DEV void rcp_hi16(uint32_t& r, uint32_t d) { asm("v_rcp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(r) : "v"(d)); }
DEV void phase_end16() {
    asm volatile("s_nop 1");
    __builtin_amdgcn_sched_barrier(0);
}
DEV void silu_l2_tile16(uint32_t (&u)[8]) {
    uint32_t e[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) e[k] = exp2neg_lo16(u[k]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 8; ++k) exp2neg_hi16(e[k], u[k]);
    phase_end16();
    const h16x2 one = {(_Float16)1.f, (_Float16)1.f};
#pragma unroll
    for (int k = 0; k < 8; ++k) e[k] = __builtin_bit_cast(uint32_t, (h16x2)(__builtin_bit_cast(h16x2, e[k]) + one));
    __builtin_amdgcn_sched_barrier(0);
    uint32_t r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = rcp_lo16(e[k]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 8; ++k) rcp_hi16(r[k], e[k]);
    phase_end16();
#pragma unroll
    for (int k = 7; k >= 0; --k) u[k] = __builtin_bit_cast(uint32_t, (h16x2)(__builtin_bit_cast(h16x2, u[k]) * __builtin_bit_cast(h16x2, r[k])));   // (last-written first)
}
template <int MODE>
__global__ void k(const unsigned* u_in, unsigned* z, int n) {       // every thread: 8 consecutive pairs
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i + 8 > n) return;
    const h2 one = {(_Float16)1.f, (_Float16)1.f};
    if (MODE == 3) {
        uint32_t u[8];
        for (int k = 0; k < 8; ++k) u[k] = u_in[i + k];
        silu_l2_tile16(u);
        for (int k = 0; k < 8; ++k) z[i + k] = u[k];
        return;
    }
    for (int k = 0; k < 8; ++k) {
        const h2 u = __builtin_bit_cast(h2, u_in[i + k]);
        h2 e, r;
        if (MODE == 0) {
            e = __builtin_elementwise_exp2(-u);
            r = one / (e + one);
        } else if (MODE == 1) {
            e = __builtin_bit_cast(h2, exp2neg_b2b(__builtin_bit_cast(unsigned, u)));
            r = __builtin_bit_cast(h2, rcp_b2b(__builtin_bit_cast(unsigned, (h2)(e + one))));
        } else {
            e = __builtin_bit_cast(h2, exp2neg_nop(__builtin_bit_cast(unsigned, u)));
            r = __builtin_bit_cast(h2, rcp_nop(__builtin_bit_cast(unsigned, (h2)(e + one))));
        }
        z[i + k] = __builtin_bit_cast(unsigned, (h2)(u * r));
    }
}
int main() {
    const int n = 1 << 20;
    std::vector<unsigned> hu(n), hz(n);
    srand(1);
    auto f2h = [](float f) { _Float16 h = (_Float16)f; unsigned short s; __builtin_memcpy(&s, &h, 2); return (unsigned)s; };
    auto h2f = [](unsigned s) { unsigned short t = (unsigned short)s; _Float16 h; __builtin_memcpy(&h, &t, 2); return (float)h; };
    for (int i = 0; i < n; ++i) {
        const float a = ((float)rand() / RAND_MAX - 0.5f) * 24.f, b = ((float)rand() / RAND_MAX - 0.5f) * 24.f;
        hu[i] = f2h(a) | (f2h(b) << 16);
    }
    unsigned *du, *dz;
    hipMalloc(&du, n * 4);
    hipMalloc(&dz, n * 4);
    hipMemcpy(du, hu.data(), n * 4, hipMemcpyHostToDevice);
    const char* names[] = {"compiler (sdwa + v_pack)", "in place, back to back", "in place, s_nop 1 between", "in place, 8 low then 8 high (dc_dev.h)"};
    int rc = 0;
    for (int mode = 0; mode < 4; ++mode) {
        hipMemset(dz, 0xff, n * 4);
        if (mode == 0) k<0><<<n / 8 / 256, 256>>>(du, dz, n);
        if (mode == 1) k<1><<<n / 8 / 256, 256>>>(du, dz, n);
        if (mode == 2) k<2><<<n / 8 / 256, 256>>>(du, dz, n);
        if (mode == 3) k<3><<<n / 8 / 256, 256>>>(du, dz, n);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", names[mode]); rc = 1; continue; }
        hipMemcpy(hz.data(), dz, n * 4, hipMemcpyDeviceToHost);
        double worst = 0.0;
        long bad = 0;
        for (int i = 0; i < n; ++i)
            for (int hlf = 0; hlf < 2; ++hlf) {
                const float u = h2f(hu[i] >> (16 * hlf)), got = h2f(hz[i] >> (16 * hlf));
                const double ref = (double)u / (1.0 + exp2(-(double)u));
                const double err = fabs(got - ref) / fmax(fabs(ref), 1e-3);
                if (!(err < 4e-3)) ++bad;
                if (err > worst) worst = err;
            }
        printf("%-40s worst relative error of u / (1 + 2^-u) %.3e, values outside 4e-3: %ld of %d\n", names[mode], worst, bad, 2 * n);
        if (bad && mode == 3) rc = 1;          // (the form the kernels use decides the exit code)
    }
    return rc;
}
