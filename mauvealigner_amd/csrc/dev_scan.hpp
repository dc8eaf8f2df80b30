// dev_scan.hpp -- tiled scans and flag compactions shared by the small device stages (chain_dev.hip, dp_batch.hip).
// The lists these stages work on (10^4 .. 10^6 entries) are far too small to fill the chip, so the pattern is
// deliberately simple: 1024 entries per workgroup (256 threads x 4 consecutive entries), per-tile aggregates written
// by a first launch, and every workgroup of the second launch sums the aggregates of the tiles before it for itself
// (a few hundred words) instead of waiting for a scan launch in between.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace devscan {

constexpr int TILE = 1024;                  // entries per workgroup: 256 threads x 4 consecutive entries

// exclusive scans over the 256 threads of a workgroup (sum, maximum with identity 0); *total = all-thread aggregate
__device__ __forceinline__ uint32_t bscan_add(uint32_t v, uint32_t *total, uint32_t *lds /*[4]*/)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    __syncthreads();                         // lds may still be read from an earlier call
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { const uint32_t c = lds[w]; if (w < wave) wbase += c; tot += c; }
    *total = tot;
    return wbase + inc - v;
}
__device__ __forceinline__ uint32_t bscan_max(uint32_t v, uint32_t *total, uint32_t *lds /*[4]*/)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o); if (lane >= o) inc = max(inc, t); }
    __syncthreads();
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { const uint32_t c = lds[w]; if (w < wave) wbase = max(wbase, c); tot = max(tot, c); }
    *total = tot;
    const uint32_t up = __shfl_up(inc, 1);
    return max(wbase, lane ? up : 0u);
}
// aggregate of the per-tile values before tile b (sum / maximum), and of all nb tiles
__device__ __forceinline__ void tiles_before_add(const uint32_t *__restrict__ v, uint32_t b, uint32_t nb, uint32_t *before, uint32_t *all, uint32_t *lds)
{
    uint32_t sb = 0, sa = 0;
    for (uint32_t t = threadIdx.x; t < nb; t += 256) { const uint32_t x = v[t]; sa += x; if (t < b) sb += x; }
    uint32_t tb, ta;
    (void)bscan_add(sb, &tb, lds); (void)bscan_add(sa, &ta, lds);
    *before = tb; *all = ta;
}

// arr[l] += v for the active lanes of a wave, which mostly share one key l (entries sorted by key): then the wave adds up
// first and issues one atomic instead of 64 on the same address.  Every lane of the wave must make the call.
__device__ __forceinline__ void wave_keyed_add(unsigned long long *arr, bool active, uint32_t l, unsigned long long v)
{
    const uint64_t act = __ballot(active);
    if (!act) return;
    const int first = __ffsll((unsigned long long)act) - 1;
    const uint32_t l0 = (uint32_t)__shfl((int)l, first);
    if (__ballot(active && l == l0) == act) {
        unsigned long long s = active ? v : 0ull;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if ((int)(threadIdx.x & 63) == first) atomicAdd(&arr[l0], s);
    } else if (active) atomicAdd(&arr[l], v);
}

// ---- flag compaction over the tiles: cmp_count (flags per tile) + cmp_write (every workgroup sums the tiles before it) ----
// F: domain(y) entries; flag(r, y); each(r, exclusive count, flag, y) for every entry; emit(r, slot, y) for the flagged;
// total(count, y) once.
template <class F>
__global__ void __launch_bounds__(256) cmp_count(F f, uint32_t *__restrict__ bcnt)
{
    __shared__ uint32_t lds[4];
    const int y = blockIdx.y;
    const uint32_t dom = f.domain(y), r0 = blockIdx.x * (uint32_t)TILE + threadIdx.x * 4u;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) if (r0 + i < dom) c += f.flag(r0 + i, y) ? 1u : 0u;
    uint32_t tc;
    (void)bscan_add(c, &tc, lds);
    if (threadIdx.x == 0) bcnt[(size_t)y * gridDim.x + blockIdx.x] = tc;
}
template <class F>
__global__ void __launch_bounds__(256) cmp_write(F f, const uint32_t *__restrict__ bcnt)
{
    __shared__ uint32_t lds[4];
    const int y = blockIdx.y;
    const uint32_t dom = f.domain(y), r0 = blockIdx.x * (uint32_t)TILE + threadIdx.x * 4u;
    uint32_t before, all;
    tiles_before_add(bcnt + (size_t)y * gridDim.x, blockIdx.x, gridDim.x, &before, &all, lds);
    bool fl[4]; uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { fl[i] = r0 + i < dom && f.flag(r0 + i, y); c += fl[i] ? 1u : 0u; }
    uint32_t dummy;
    uint32_t o = before + bscan_add(c, &dummy, lds);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (r0 + i >= dom) break;
        f.each(r0 + i, o, fl[i], y);
        if (fl[i]) { f.emit(r0 + i, o, y); o++; }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) f.total(all, y);
}


// ---- exclusive scan of values: vscan_partial (tile sums) + vscan_write (out[i] = sum of in[0..i), *total = sum) ----
// In: a functor value(i) -> T so that the input need not exist as an array.
template <class T, class In>
__global__ void __launch_bounds__(256) vscan_partial(In in, uint32_t n, T *__restrict__ bsum)
{
    __shared__ T red[4];
    const uint32_t r0 = blockIdx.x * (uint32_t)TILE + threadIdx.x * 4u;
    T s = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) if (r0 + i < n) s += in.value(r0 + i);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
template <class T, class In>
__global__ void __launch_bounds__(256) vscan_write(In in, uint32_t n, const T *__restrict__ bsum, T *__restrict__ out, T *__restrict__ total)
{
    __shared__ T red[4];
    __shared__ T s_before, s_all;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // tiles before this one, and all of them
    T sb = 0, sa = 0;
    for (uint32_t t = threadIdx.x; t < gridDim.x; t += 256) { const T x = bsum[t]; sa += x; if (t < blockIdx.x) sb += x; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sb += __shfl_down(sb, o); sa += __shfl_down(sa, o); }
    if (lane == 0) red[wave] = sb;
    __syncthreads();
    if (threadIdx.x == 0) s_before = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    if (lane == 0) red[wave] = sa;
    __syncthreads();
    if (threadIdx.x == 0) s_all = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    const uint32_t r0 = blockIdx.x * (uint32_t)TILE + threadIdx.x * 4u;
    T v[4], s = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { v[i] = r0 + i < n ? in.value(r0 + i) : (T)0; s += v[i]; }
    T inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const T t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    __syncthreads();
    if (lane == 63) red[wave] = inc;
    __syncthreads();
    T run = s_before + inc - s;
    for (int w = 0; w < wave; w++) run += red[w];
#pragma unroll
    for (int i = 0; i < 4; i++) { if (r0 + i < n) out[r0 + i] = run; run += v[i]; }
    if (blockIdx.x == 0 && threadIdx.x == 0) { if (total) *total = s_all; out[n] = s_all; }
}

}  // namespace devscan
