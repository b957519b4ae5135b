// dev_common.h -- device helpers shared by the kernels of libcutadapt_hip.so (kernels.hip, multi.hip):
// wave constants, the packed read layout, work dequeue and the 16-characters-per-load chunk reader.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WAVE 64
__device__ __forceinline__ int wave_lane() { return threadIdx.x & (WAVE - 1); }
// (threadIdx.x >> 6 is wave-uniform, but the compiler does not know: values derived from it would live in VGPRs and
// buffer resources built from them would be looped over lane by lane)
__device__ __forceinline__ int wave_index() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// Where does read `r` live?  (packed layout, see include/cutadapt_hip.h)
__device__ __forceinline__ void read_extent(const int64_t* offsets, const int32_t* lens, int64_t r,
                                            int64_t& off, int64_t& n) {
    off = offsets[r];
    n = lens ? (int64_t)lens[r] : offsets[r + 1] - off;
}

// ... or, for a batch of equally long reads the host knows about (ulen > 0), at ufirst + r * ulen: no offsets array
__device__ __forceinline__ void read_extent(const int64_t* offsets, const int32_t* lens, const int64_t ufirst,
                                            const int32_t ulen, int64_t r, int64_t& off, int64_t& n) {
    if (ulen > 0) { off = ufirst + r * (int64_t)ulen; n = ulen; }
    else read_extent(offsets, lens, r, off, n);
}

// ---------------------------------------------------------------------------------------------
// Work distribution: waves pull chunks of 64 work items from a device counter ("dequeue",
// the cheapest cross-CU primitive on this chip) so that long and short reads balance.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t wave_dequeue(unsigned long long* counter, unsigned items = WAVE) {
    unsigned long long base = 0;
    if (wave_lane() == 0) base = atomicAdd(counter, (unsigned long long)items);
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)base);
    unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
    return (int64_t)(((unsigned long long)hi << 32) | lo);
}


// ---------------------------------------------------------------------------------------------
// Read bytes: 16 characters per lane per global load, held in four VGPRs.
// Reads are packed back to back, so a read starts at an arbitrary byte; gfx950 executes
// unaligned global_load_dwordx4.  load_chunk never touches memory outside q[0, n):
//   interior chunk  -> one 16-byte load at q + pos
//   last chunk      -> the 16 bytes that END at the read end, funnel-shifted down
//   reads < 16 bytes-> assembled from byte loads
// Characters at positions >= limit come back as NUL (which matches nothing in any table).
// ---------------------------------------------------------------------------------------------
struct Chunk { unsigned w[4]; };

struct __attribute__((packed, aligned(1))) Unaligned16 { unsigned w[4]; };

__device__ __forceinline__ Chunk load_chunk(const uint8_t* q, int pos, int n, int limit) {
    Chunk c;
    c.w[0] = c.w[1] = c.w[2] = c.w[3] = 0;
    if (pos >= limit) return c;
    if (pos + 16 <= n) {
        Unaligned16 u;
        __builtin_memcpy(&u, q + pos, 16);
        c.w[0] = u.w[0]; c.w[1] = u.w[1]; c.w[2] = u.w[2]; c.w[3] = u.w[3];
    } else if (n >= 16) {
        Unaligned16 u;
        __builtin_memcpy(&u, q + (n - 16), 16);
        const int s = pos - (n - 16);                     // 1..15 bytes to drop
        const int dw = s >> 2, sh = (s & 3) * 8;
        unsigned x0 = u.w[0], x1 = u.w[1], x2 = u.w[2], x3 = u.w[3];
        if (dw >= 2) { x0 = x2; x1 = x3; x2 = 0; x3 = 0; }
        if (dw & 1) { x0 = x1; x1 = x2; x2 = x3; x3 = 0; }
        c.w[0] = (unsigned)((((unsigned long long)x1 << 32) | x0) >> sh);
        c.w[1] = (unsigned)((((unsigned long long)x2 << 32) | x1) >> sh);
        c.w[2] = (unsigned)((((unsigned long long)x3 << 32) | x2) >> sh);
        c.w[3] = x3 >> sh;
    } else {
        // read shorter than 16 bytes: push its bytes in from the top, last byte first (a rolled
        // loop over four named registers: no dynamic register indexing, tiny register footprint)
        unsigned x0 = 0, x1 = 0, x2 = 0, x3 = 0;
#pragma unroll 1
        for (int t = n - 1; t >= pos; --t) {
            x3 = (x3 << 8) | (x2 >> 24);
            x2 = (x2 << 8) | (x1 >> 24);
            x1 = (x1 << 8) | (x0 >> 24);
            x0 = (x0 << 8) | (unsigned)q[t];
        }
        c.w[0] = x0; c.w[1] = x1; c.w[2] = x2; c.w[3] = x3;
    }
    const int keep = limit - pos;                         // characters of this chunk inside [pos, limit)
    if (keep < 16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int v = keep - 4 * i;                   // valid bytes in dword i
            const unsigned msk = v >= 4 ? 0xFFFFFFFFu : (v <= 0 ? 0u : ((1u << (8 * v)) - 1u));
            c.w[i] &= msk;
        }
    }
    return c;
}

// The chunk at FRAME position `pos` of a view that is streamed END-ALIGNED in a frame of n characters (pad = n - view
// length NULs in front; qv: the view's first character): frame position j is the view's j - pad.  pad == 0: load_chunk.
__device__ __forceinline__ Chunk load_chunk_frame(const uint8_t* qv, const int pos, const int n, const int pad, const int limit) {
    if (pad == 0) return load_chunk(qv, pos, n, limit);
    const int vp = pos - pad, nv = n - pad;
    Chunk c;
    c.w[0] = c.w[1] = c.w[2] = c.w[3] = 0;
    if (pos >= limit || vp + 16 <= 0 || nv <= 0) return c;
    if (vp >= 0) return load_chunk(qv, vp, nv, limit - pad);
    // the chunk straddles the view's first character: the view's first 16 + vp characters, moved up by -vp bytes
    const Chunk lo = load_chunk(qv, 0, nv, min(limit - pad, 16 + vp));
    const int s = -vp;                                                  // 1 .. 15
    const int dw = s >> 2, sh = (s & 3) * 8;
    unsigned x0 = lo.w[0], x1 = lo.w[1], x2 = lo.w[2], x3 = lo.w[3];
    if (dw >= 2) { x3 = x1; x2 = x0; x1 = 0; x0 = 0; }
    if (dw & 1) { x3 = x2; x2 = x1; x1 = x0; x0 = 0; }
    c.w[0] = x0 << sh;
    c.w[1] = (unsigned)((((unsigned long long)x1 << 32) | x0) >> (32 - sh) >> 0);
    c.w[2] = (unsigned)((((unsigned long long)x2 << 32) | x1) >> (32 - sh));
    c.w[3] = (unsigned)((((unsigned long long)x3 << 32) | x2) >> (32 - sh));
    if (sh == 0) { c.w[1] = x1; c.w[2] = x2; c.w[3] = x3; }
    return c;
}

// Frame characters fc .. fc + 15 of a view of `len` characters that ends at byte `veb` of seqs, END-ALIGNED in a frame of n
// characters (the gathering copy of the streaming prefilters' RV forms: views anywhere in the buffer).  Bytes safe_lo ..
// safe_hi - 1 are known to lie inside the buffer (the extent of the piece's own views): a unit inside that range is ONE
// 16-byte load, whatever it holds outside its view is masked by the caller (skip in front, the frame's end behind).  A unit
// that reaches out of the range -- the first view's first unit, the last view's last -- is the 16 bytes at the view's edge,
// shifted; views under 16 characters byte by byte.  Nothing outside [safe_lo, safe_hi) is touched.
__device__ __forceinline__ void gather_frame_unit(const uint8_t* seqs, const int64_t veb, const int len, const int n, const int fc,
                                                  const int64_t safe_lo, const int64_t safe_hi,
                                                  unsigned& x0, unsigned& x1, unsigned& x2, unsigned& x3) {
    const int64_t fa = veb - n + fc;
    if (fa >= safe_lo && fa + 16 <= safe_hi) {
        Unaligned16 v;
        __builtin_memcpy(&v, seqs + fa, 16);
        x0 = v.w[0]; x1 = v.w[1]; x2 = v.w[2]; x3 = v.w[3];
        return;
    }
    const int64_t vsb = veb - len;
    x0 = x1 = x2 = x3 = 0;
    if (len >= 16) {
        int64_t la = fa < vsb ? vsb : fa;
        la = la > veb - 16 ? veb - 16 : la;
        Unaligned16 v;
        __builtin_memcpy(&v, seqs + la, 16);
        x0 = v.w[0]; x1 = v.w[1]; x2 = v.w[2]; x3 = v.w[3];
        const int sft = (int)(fa - la);                                  // > 0: the unit reaches behind the view; < 0: in front of it
        if (sft > 0) {
            const int dw = sft >> 2, sh = (sft & 3) * 8;
            if (dw >= 2) { x0 = x2; x1 = x3; x2 = 0; x3 = 0; }
            if (dw & 1) { x0 = x1; x1 = x2; x2 = x3; x3 = 0; }
            const unsigned y0 = (unsigned)((((unsigned long long)x1 << 32) | x0) >> sh);
            const unsigned y1 = (unsigned)((((unsigned long long)x2 << 32) | x1) >> sh);
            const unsigned y2 = (unsigned)((((unsigned long long)x3 << 32) | x2) >> sh);
            x3 = x3 >> sh; x0 = y0; x1 = y1; x2 = y2;
        } else if (sft < 0) {
            const int up = -sft, dw = up >> 2, sh = (up & 3) * 8;
            if (dw >= 2) { x3 = x1; x2 = x0; x1 = 0; x0 = 0; }
            if (dw & 1) { x3 = x2; x2 = x1; x1 = x0; x0 = 0; }
            const unsigned y3 = (unsigned)((((unsigned long long)x3 << 32) | x2) >> (32 - sh));
            const unsigned y2 = (unsigned)((((unsigned long long)x2 << 32) | x1) >> (32 - sh));
            const unsigned y1 = (unsigned)((((unsigned long long)x1 << 32) | x0) >> (32 - sh));
            if (sh) { x3 = y3; x2 = y2; x1 = y1; x0 = x0 << sh; }
        }
    } else {
        unsigned x[4] = {0u, 0u, 0u, 0u};
#pragma unroll 1
        for (int b = 0; b < 16; ++b) {
            const int64_t at = fa + b;
            if (at >= vsb && at < veb) x[b >> 2] |= (unsigned)seqs[at] << (8 * (b & 3));
        }
        x0 = x[0]; x1 = x[1]; x2 = x[2]; x3 = x[3];
    }
}
// wave-wide minimum / maximum of a 32-bit value (every lane takes part)
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) v = min(v, __shfl_xor(v, sft, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) v = max(v, __shfl_xor(v, sft, 64));
    return v;
}

// The chunk at `pos` of lanes whose chunk lies inside their read (pos + 16 <= n): ONE load, nothing else -- for callers
// that have asked the wave first; load_chunk's three cases and the masks behind them are ~35 instructions per call.
__device__ __forceinline__ Chunk load_chunk_interior(const uint8_t* q, const int pos, const bool want) {
    Chunk c;
    c.w[0] = c.w[1] = c.w[2] = c.w[3] = 0;
    if (want) {
        Unaligned16 u;
        __builtin_memcpy(&u, q + pos, 16);
        c.w[0] = u.w[0]; c.w[1] = u.w[1]; c.w[2] = u.w[2]; c.w[3] = u.w[3];
    }
    return c;
}

// byte t (0..15) of a chunk; t is wave-uniform or a compile-time constant
__device__ __forceinline__ unsigned chunk_byte(const Chunk& c, int t) {
    const int d = t >> 2;
    const unsigned w = d == 0 ? c.w[0] : (d == 1 ? c.w[1] : (d == 2 ? c.w[2] : c.w[3]));
    return (w >> ((t & 3) * 8)) & 0xFFu;
}


// One read's result row, status and best adapter (merge_best: 0 = plain store, 1 = MultipleAdapters' merge with what an
// earlier adapter of the plan left, reference adapters.py:1278-1285, 2 = the plan's first adapter on rows the entry point
// has zeroed).  Shared by the cost scans (kernels.hip) and the cell DP kernels.
__device__ __forceinline__ void store_result(int32_t* out6, uint8_t* status, int32_t* best_adapter,
                                             const int adapter_index, const int merge_best, const int64_t r,
                                             const bool invalid, const bool found, const int t0, const int t1,
                                             const int t2, const int t3, const int score, const int cost) {
    int32_t* o = out6 + r * 6;
    if (merge_best == 2) {
        // the plan's FIRST adapter on rows the entry point has zeroed (status 0, tuple 0, best_adapter -1): nothing
        // to compare with, nothing to read -- a read without a match keeps its zeros
        if (invalid) {
            status[r] = 2;
        } else if (found) {
            o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = score; o[5] = cost;
            status[r] = 1;
            if (best_adapter) best_adapter[r] = adapter_index;
        }
    } else if (merge_best) {
        // MultipleAdapters.match_to (adapters.py:1278-1285), see k_dp
        if (invalid) {
            status[r] = 2;
        } else if (found) {
            const bool had = status[r] == 1;
            if (status[r] != 2 && (!had || score > o[4] || (score == o[4] && cost < o[5]))) {
                o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = score; o[5] = cost;
                status[r] = 1;
                if (best_adapter) best_adapter[r] = adapter_index;
            }
        }
    } else {
        status[r] = invalid ? (uint8_t)2 : (found ? (uint8_t)1 : (uint8_t)0);
        if (found && !invalid) { o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = score; o[5] = cost; }
        else { o[0] = o[1] = o[2] = o[3] = o[4] = o[5] = 0; }
    }
}
