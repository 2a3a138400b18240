// How do two waves of a SIMD share its issue slots?  (round 4: profiles/r04/wave_timeline.txt)
// One block of 512 threads per CU = two waves on each of the four SIMDs (256 VGPRs per wave pin that occupancy), every wave runs
// the same loop of dependent-free v_mad_u64_u32 / v_addc pairs; each wave records wall_clock64 at its start and end and its
// wave slot.  Modes: 0 plain; 1 slot 1 raises its priority (s_setprio 3) for its whole life; 2 both waves alternate their
// priority every `period` loop iterations, in opposite phase (slot s: ((it / period) + s) & 1); 3 time slices of the 100 MHz
// counter, opposite phase by slot.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void __launch_bounds__(512, 1) k_arb(int iters, int mode, int period, unsigned long long *out) {
    const unsigned slot = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (3 << 11)) & 1u;
    const unsigned long long t0 = wall_clock64();
    if (mode == 1 && slot == 1) __builtin_amdgcn_s_setprio(3);
    unsigned x = threadIdx.x, y;
    for (int it = 0; it < iters; it++) {
        if (mode == 2) {
            if ((((unsigned)it / (unsigned)period) + slot) & 1u) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        } else if (mode == 3) {
            const unsigned turn = (unsigned)(__builtin_amdgcn_s_memrealtime() >> period);
            if ((turn ^ slot) & 1u) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        asm volatile(
            ".rept 64\n\t"
            "v_mad_u64_u32 v[20:21], s[40:41], %1, %1, v[20:21]\n\t"
            "v_mad_u64_u32 v[22:23], s[42:43], %1, %1, v[22:23]\n\t"
            "v_mad_u64_u32 v[24:25], s[44:45], %1, %1, v[24:25]\n\t"
            "v_addc_co_u32 v36, s[40:41], 0, v36, s[40:41]\n\t"
            "v_addc_co_u32 v37, s[42:43], 0, v37, s[42:43]\n\t"
            "v_addc_co_u32 v38, s[44:45], 0, v38, s[44:45]\n\t"
            ".endr\n\t"
            "v_xor_b32 %0, v20, v36"
            : "=v"(y) : "v"(x)
            : "v20", "v21", "v22", "v23", "v24", "v25", "v36", "v37", "v38", "s40", "s41", "s42", "s43", "s44", "s45", "v255");
        x += y & 1u;
    }
    const unsigned long long t1 = wall_clock64();
    if ((threadIdx.x & 63u) == 0) {
        const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64;
        out[3 * w] = t0;
        out[3 * w + 1] = t1;
        out[3 * w + 2] = slot | ((unsigned long long)(x & 1u) << 32) | ((unsigned long long)(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7u) << 8);
    }
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount;
    const size_t nw = (size_t)blocks * 8;
    unsigned long long *d;
    hipMalloc(&d, 3 * nw * sizeof(unsigned long long));
    std::vector<unsigned long long> h(3 * nw);
    struct { int mode, period; const char *name; } runs[] = {
        {0, 0, "plain"}, {1, 0, "slot 1 at s_setprio 3"}, {2, 1, "alternate every iteration"}, {2, 8, "alternate every 8 iterations"},
        {2, 64, "alternate every 64 iterations"}, {3, 10, "time slices of 10 us"}, {3, 13, "time slices of 82 us"}, {0, 0, "plain again"}};
    for (auto &r : runs) {
        hipLaunchKernelGGL(k_arb, dim3(blocks), dim3(512), 0, 0, iters, r.mode, r.period, d);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double dur[2] = {0, 0};
        int cnt[2] = {0, 0};
        unsigned long long tmin = ~0ull, tmax = 0;
        for (size_t w = 0; w < nw; w++) {
            const int s = (int)(h[3 * w + 2] & 1u);
            dur[s] += (double)(h[3 * w + 1] - h[3 * w]) / 100.0;
            cnt[s]++;
            if (h[3 * w] < tmin) tmin = h[3 * w];
            if (h[3 * w + 1] > tmax) tmax = h[3 * w + 1];
        }
        std::printf("%-34s slot 0: %8.1f us (%d waves)   slot 1: %8.1f us (%d waves)   kernel span %8.1f us\n", r.name,
                    dur[0] / (cnt[0] ? cnt[0] : 1), cnt[0], dur[1] / (cnt[1] ? cnt[1] : 1), cnt[1], (double)(tmax - tmin) / 100.0);
        if (r.mode == 0) {      // per XCD: the mean life of a PAIR of waves (slot 0 + slot 1: the SIMD's work), pure arithmetic
            double sum[8] = {0};
            int c8[8] = {0};
            for (size_t w = 0; w < nw; w++) {
                const int xcc = (int)((h[3 * w + 2] >> 8) & 7u);
                sum[xcc] += (double)(h[3 * w + 1] - h[3 * w]) / 100.0;
                c8[xcc]++;
            }
            std::printf("    per XCC, mean wave life (us):");
            for (int x = 0; x < 8; x++) std::printf(" %d: %.1f", x, c8[x] ? sum[x] / c8[x] : 0.0);
            std::printf("\n");
        }
    }
    return 0;
}
