// filter_common.h -- device helpers shared by the prefilter kernels (kernels.hip, stream2.hip): clearing the result
// rows on the way through the batch, and the tile's survivor staging with its key-ordered flush.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "dev_common.h"

// Zero the result rows (24 bytes each) of the `cnt` (<= 64, wave-uniform) consecutive reads from `base` on: the
// rows are contiguous, every store instruction of the wave writes 512 contiguous bytes.
__device__ __forceinline__ void clear_rows(int32_t* out6, int32_t* best, const int64_t base, const int cnt, const int lane) {
    if (best && lane < cnt) best[base + lane] = -1;
    int32_t* const o = out6 + base * 6;
    if ((reinterpret_cast<uintptr_t>(o) & 7u) == 0) {
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int u = lane + WAVE * k;                                  // 8-byte unit
            if (u < cnt * 3) *reinterpret_cast<u32x2*>(o + 2 * u) = (u32x2)(0u);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int u = lane + WAVE * k;
            if (u < cnt * 6) o[u] = 0;
        }
    }
}

// (Args: FilterArgs, or any struct with the members present / status / queue / queue_keys / queue_count.)
// The survivors of a tile, staged in LDS with their keys, leave as one key-ordered run: exclusive scan of the
// 256-bin histogram (one thread per bin), one atomic for the run, counting sort into the global queue.
// blockDim.x >= 256 == CAH_QUEUE_BINS (a multiple of 64).  s_scratch: 8 words.
template <class Args>
__device__ __forceinline__ void flush_tile_queue(const Args& a, int64_t tile_base, const uint16_t* s_idx,
                                                 const uint8_t* s_key, unsigned* s_hist, unsigned* s_cursor,
                                                 const unsigned count, unsigned* s_scratch,
                                                 unsigned long long& s_qbase) {
    const int lane = wave_lane(), wave = threadIdx.x >> 6;
    const bool bin = threadIdx.x < CAH_QUEUE_BINS;
    const unsigned c = bin ? s_hist[threadIdx.x] : 0u;
    unsigned incl = c;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const unsigned o = __shfl_up(incl, d, WAVE);
        if (lane >= d) incl += o;
    }
    if (bin && lane == WAVE - 1) s_scratch[wave] = incl;
    if (threadIdx.x == 0) s_qbase = count ? atomicAdd(a.queue_count, (unsigned long long)count) : 0ull;
    __syncthreads();
    if (bin) {
        unsigned before = 0;
#pragma unroll
        for (int w = 0; w < 3; ++w) if (w < wave) before += s_scratch[w];
        s_hist[threadIdx.x] = before + incl - c;
    }
    __syncthreads();
    const unsigned long long qbase = s_qbase;
    for (unsigned e = threadIdx.x; e < count; e += blockDim.x) {
        const unsigned key = s_key[e];
        const unsigned p = s_hist[key] + atomicAdd(&s_cursor[key], 1u);
        a.queue[qbase + p] = (int32_t)(tile_base + s_idx[e]);
        a.queue_keys[qbase + p] = (uint8_t)key;
    }
}

// a read's verdict: present[] (mode 0: a.present is set) or a slot of the tile's survivor staging (mode 1)
template <class Args>
__device__ __forceinline__ void lean_emit(const Args& a, int64_t r, int64_t tile_base, bool valid, bool hit,
                                          bool invalid, int hit_pos, uint16_t* s_idx, uint8_t* s_key,
                                          unsigned* s_hist, unsigned& s_count) {
    if (a.present) {
        if (valid) a.present[r] = invalid ? (uint8_t)2 : (hit ? (uint8_t)1 : (uint8_t)0);
    } else {
        if (valid && invalid) a.status[r] = 2;
        const bool push = valid && hit && !invalid;
        const unsigned long long bal = __ballot(push);
        if (bal) {
            const int lane = wave_lane();
            unsigned slot = 0;
            if (lane == 0) slot = atomicAdd(&s_count, (unsigned)__popcll(bal));
            slot = __builtin_amdgcn_readfirstlane(slot);
            if (push) {
                const int e = (int)slot + __popcll(bal & ((1ull << lane) - 1ull));
                const int key = min(hit_pos >> CAH_KEY_SHIFT, CAH_QUEUE_BINS - 1);
                s_idx[e] = (uint16_t)(r - tile_base);
                s_key[e] = (uint8_t)key;
                atomicAdd(&s_hist[key], 1u);
            }
        }
    }
}

