// probe_gap.hip - what does the gap between two DEPENDENT kernel launches depend on?  (k_layer: 7 us between the last workgroup's end
// and the next launch's first begin, 2.5 us of start skew - 18 % of its launch period; profiles/r04_diag_launch_gap.txt.)
// Each workgroup stamps s_memrealtime at its begin and end, busy-waits `spin_us` in between and optionally streams bytes; 24
// launches back to back on one stream; reported: median over launch pairs of (first begin of launch i+1) - (last end of launch i),
// and of the start skew (last begin - first begin).   build: hipcc --offload-arch=gfx950 -O3 tools/probe_gap.hip -o tools/probe_gap
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

__global__ void k_probe(unsigned long long* stamps, int launch, int spin_ticks, float* buf, int floats_per_wg, int rw) {
    extern __shared__ char lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) stamps[((size_t)launch * gridDim.x + blockIdx.x) * 2] = t0;
    if (floats_per_wg) {
        float* p = buf + (size_t)blockIdx.x * floats_per_wg;
        float acc = 0.f;
        for (int i = threadIdx.x; i < floats_per_wg; i += blockDim.x) {
            if (rw & 1) acc += p[i];
            if (rw & 2) p[i] = acc + (float)launch;
        }
        if (acc == 123.456f) lds[0] = 1;
    }
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(8);
    if (lds != nullptr && threadIdx.x == 5000) lds[7] = 2;
    __syncthreads();
    if (threadIdx.x == 0) stamps[((size_t)launch * gridDim.x + blockIdx.x) * 2 + 1] = __builtin_amdgcn_s_memrealtime();
}

int main() {
    const int NL = 24;
    unsigned long long* d_st;
    float* d_buf;
    hipMalloc(&d_st, (size_t)NL * 1024 * 2 * 8);
    hipMalloc(&d_buf, (size_t)1024 * 65536 * 4);
    hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipStream_t st;
    hipStreamCreate(&st);
    struct Cfg { int grid, block, lds, spin_us, kib_per_wg, rw; const char* what; };
    const Cfg cfgs[] = {
        {256, 64, 0, 5, 0, 0, "trivial: 256 x 64, no LDS"},
        {228, 512, 0, 20, 0, 0, "228 x 512, no LDS"},
        {228, 512, 64 * 1024, 20, 0, 0, "228 x 512, 64 KiB LDS"},
        {228, 512, 152 * 1024, 20, 0, 0, "228 x 512, 152 KiB LDS"},
        {228, 512, 152 * 1024, 40, 0, 0, "228 x 512, 152 KiB LDS, 40 us"},
        {228, 256, 152 * 1024, 20, 0, 0, "228 x 256, 152 KiB LDS"},
        {228, 512, 152 * 1024, 20, 128, 1, "... + 128 KiB read per workgroup"},
        {228, 512, 152 * 1024, 20, 128, 2, "... + 128 KiB written per workgroup"},
        {228, 512, 152 * 1024, 20, 128, 3, "... + 128 KiB read and written"},
        {456, 512, 76 * 1024, 20, 0, 0, "456 x 512, 76 KiB LDS (two per CU)"},
        {29, 256, 110 * 1024, 20, 0, 0, "29 x 256, 110 KiB LDS (k_layer16 at bs = 1)"},
    };
    for (int graph = 0; graph < 2; ++graph)
        for (const Cfg& c : cfgs) {
            hipMemsetAsync(d_st, 0, (size_t)NL * 1024 * 2 * 8, st);
            hipGraph_t g = nullptr;
            hipGraphExec_t ge = nullptr;
            if (graph) hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
            for (int l = 0; l < NL; ++l)
                k_probe<<<dim3(c.grid), dim3(c.block), c.lds, st>>>(d_st, l, c.spin_us * 100, d_buf, c.kib_per_wg * 256, c.rw);
            if (graph) {
                hipStreamEndCapture(st, &g);
                hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
                hipGraphLaunch(ge, st);
            }
            hipStreamSynchronize(st);
            std::vector<unsigned long long> h((size_t)NL * c.grid * 2);
            hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost);
            std::vector<double> gaps, skews, tails;
            for (int l = 4; l + 1 < NL; ++l) {
                unsigned long long fb = ~0ull, lb = 0, le = 0, fe = ~0ull, nb = ~0ull;
                for (int w = 0; w < c.grid; ++w) {
                    fb = std::min(fb, h[((size_t)l * c.grid + w) * 2]);
                    lb = std::max(lb, h[((size_t)l * c.grid + w) * 2]);
                    le = std::max(le, h[((size_t)l * c.grid + w) * 2 + 1]);
                    fe = std::min(fe, h[((size_t)l * c.grid + w) * 2 + 1]);
                    nb = std::min(nb, h[((size_t)(l + 1) * c.grid + w) * 2]);
                }
                gaps.push_back(((double)nb - (double)le) / 100.0);
                skews.push_back(((double)lb - (double)fb) / 100.0);
                tails.push_back(((double)le - (double)fe) / 100.0);
            }
            auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
            printf("%-6s %-48s gap %6.2f us   start skew %6.2f us   end spread %6.2f us\n", graph ? "graph" : "eager", c.what, med(gaps), med(skews), med(tails));
            if (ge) hipGraphExecDestroy(ge);
            if (g) hipGraphDestroy(g);
        }
    return 0;
}
