// qp/sched.hpp -- interior-point vectors (IpmVec) and the work ordering (sched_*: instances that ran the QP loop in the previous solve are handed out first).
// Part of ONE translation unit: qp_kernel.hip includes these headers in layer order (tiles -> sweeps -> window -> sched -> qp_body ->
// lin_phase -> fused -> windowed -> pit) and instantiates the kernels between them; see the file map at the head of qp_kernel.hip.
#pragma once

namespace brov {

// one interior-point vector.  MODE 1 (fused kernels, nv <= 128): two elements per lane, in registers for the whole loop.
// MODE 0 (streaming kernel): an HBM array, read and written element by element.  MODE 2 (windowed kernel): an HBM array with a
// register copy of the lane's T elements that lives for one group of element loops -- fetch() at the head of the group (all the
// group's loads are requested back to back, ahead of its first store: written element by element the compiler has to keep every
// load behind the previous element's stores, which may alias, and the single resident wave then sits through one L2 / HBM
// round trip per element instead of one per group), flush() at its end.  Between groups (across the sweeps) only HBM holds it.
template <int MODE, int T>
struct IpmVec {
    double r[T];
    double* g;
    __device__ __forceinline__ double get(int t, int j) const { return MODE ? r[t % T] : g[j]; }
    __device__ __forceinline__ void set(int t, int j, double v) { if (MODE) r[t % T] = v; else g[j] = v; }
    // element indices are UNSIGNED: base pointer (uniform, SGPR pair) + zero-extended 32-bit offset is one addressing mode of
    // global_load / global_store, so the 8 offsets of a lane serve every vector; with a signed index the compiler forms one 64-bit
    // address per element and vector (160 VGPRs in the windowed kernel) and keeps them all live across the interior-point loop
    __device__ __forceinline__ void fetch(int lane, int nv) {
        if constexpr (MODE == 2) {
#pragma unroll
            for (int t = 0; t < T; t++) { const unsigned j = (unsigned)lane + 64u * t; r[t] = g[j < (unsigned)nv ? j : 0u]; }
        }
    }
    __device__ __forceinline__ void flush(int lane, int nv) const {
        if constexpr (MODE == 2) {
#pragma unroll
            for (int t = 0; t < T; t++) { const unsigned j = (unsigned)lane + 64u * t; if (j < (unsigned)nv) g[j] = r[t]; }
        }
    }
};

// ---- work ordering: expensive instances first ----------------------------------------------------------------------------------
// A launch ends with its slowest instance, and which instances are slow is known in advance with good odds: an instance whose QP had
// active bounds in the previous control tick (it ran active-set tries / interior-point iterations: 2 .. 6 times the cycles of an
// early exit) almost always has them again in this one.  Every solve therefore records the instances that entered the QP loop
// (atomic append to a list L of length n, and pos[b] = position in L or -1), and the next solve hands THOSE out first:
//     (per class of instances, see below)
//     ticket t <  n            -> instance L[t]
//     ticket t >= n, pos[t] < 0 -> instance t
//     ticket t >= n, pos[t] >= 0 (t is in L): the prefix instance its list position names, following pos while that instance is
//                                 itself in L -- the chain ends on a prefix instance outside L, and two chains never meet (pos is
//                                 injective on L), so the map is a bijection of [0, B)
// Tickets are block indices (fused / streaming kernels: the hardware dispatches blocks in index order) or the atomic counter's
// values (windowed kernel).  Only the ORDER of the work changes: every instance is still solved by one wave on its own data, the
// results are bit-identical with and without (tests/test_gpu_edge.py).  Three buffers rotate: read (written by the previous
// solve), written, and zeroed for the next solve.  Measured: mixed batch 17.4 -> see DESIGN.md section 7.
// Contention: all resident waves reach the end of an equally long solve within microseconds of each other, and atomics on ONE
// address serialise (~4 ns each: 4 us per round of 1024 waves when every instance runs the loop, 7 % of that leg).  The instances
// are therefore split into 64 classes (index mod 64), each with its own counter (on its own 128-byte line), list and bijection;
// ticket t is served by class t mod 64, position t / 64.
constexpr int kSchedClasses = 64, kSchedCntStride = 32;
__host__ __device__ inline int sched_class_len(int B) { return (B + kSchedClasses - 1) / kSchedClasses; }
__host__ __device__ inline int sched_buffer_ints(int B) { return kSchedClasses * kSchedCntStride + kSchedClasses * sched_class_len(B) + B; }
__device__ __forceinline__ int sched_map(const DevParams& P, int t) {
    if (!P.sched) return t;
    const int32_t* __restrict__ Rd = P.sched + (size_t)P.sched_r * P.sched_stride;
    const int k = t & (kSchedClasses - 1), i = t >> 6, Bc = sched_class_len(P.B);
    const int n = Rd[k * kSchedCntStride];
    // nothing to gain when most instances of the class are listed (every ticket would pay a dependent look-up for an order that does
    // not matter)
    if (n <= 0 || 2 * n > Bc) return t;
    const int32_t* __restrict__ L = Rd + kSchedClasses * kSchedCntStride + k * Bc;
    const int32_t* __restrict__ pos = Rd + kSchedClasses * kSchedCntStride + kSchedClasses * Bc;
    if (i < n) return L[i];
    int x = pos[t];
    if (x < 0) return t;
    for (int guard = 0; guard < n; guard++) {
        const int y = pos[x * kSchedClasses + k];
        if (y < 0) break;
        x = y;
    }
    return x * kSchedClasses + k;
}
// Ticket and note are WAVE-UNIFORM (round 4).  Round 3 took the ticket under `if (lane == 0)` -- a divergent region ahead of the QP
// loop, next to the place where hipcc's register allocator once put AGPR copies of live registers ahead of the exec restore of a
// join block (scripts/check_exec_restore.py).  Now the atomic is ONE inline-assembly block that narrows exec to lane 0 and restores
// it itself: the compiler sees straight-line code and builds no join block here.  The block waits for the returned value (an
// inline-asm result the compiler might otherwise copy before it has landed): one L2 round trip, ~1.5 us, per instance that runs
// the QP loop (>= 60 us).  Pointers are forced into SGPRs, the two stores of sched_note are issued by all lanes with identical
// address and data.  What round 4 learned about the defect itself: it is NOT tied to this region.  Taking the ticket at the end of
// the wave instead (BROV_SCHED_TICKET_LATE) moves the allocator's copies to the join block of a guarded store of the first-guess
// loop, 40 lines away -- any of the kernel's ~1000 divergent regions can host it when the allocation shifts, which is why the
// link rule runs the checker on every build and tests/test_kernel_resources.py keeps that statement order as a live canary.
__device__ __forceinline__ const int32_t* uniform_ptr(const int32_t* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const int32_t*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int wave_atomic_inc(int32_t* addr_uniform) {
    int ret = 0;
    const int zero = 0, one = 1;
    unsigned long long saved;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, 1\n\t"
        "global_atomic_add %[r], %[off], %[one], %[base] sc0\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "s_mov_b64 exec, %[sv]"
        : [r] "+v"(ret), [sv] "=&s"(saved)
        : [off] "v"(zero), [one] "v"(one), [base] "s"(addr_uniform)
        : "memory");
    return __builtin_amdgcn_readfirstlane(ret);
}
// sched_ticket: where the instance enters the QP loop (wave-uniform result); sched_note: at the end of the wave, all lanes storing
// identical data to identical addresses
__device__ __forceinline__ int sched_ticket(const DevParams& P, int b) {
    if (!P.sched) return -1;
    int32_t* Wr = (int32_t*)uniform_ptr(P.sched + (size_t)P.sched_w * P.sched_stride);
    return wave_atomic_inc(Wr + (b & (kSchedClasses - 1)) * kSchedCntStride);
}
__device__ __forceinline__ void sched_note(const DevParams& P, int b, int p) {
    if (!P.sched) return;
    int32_t* Wr = (int32_t*)uniform_ptr(P.sched + (size_t)P.sched_w * P.sched_stride);
    const int Bc = sched_class_len(P.B), k = b & (kSchedClasses - 1);
    if (p >= 0 && p < Bc) Wr[kSchedClasses * kSchedCntStride + k * Bc + p] = b;
    Wr[kSchedClasses * kSchedCntStride + kSchedClasses * Bc + b] = p;
}
// did instance b run the QP loop in the previous solve?  (pos[b] of the buffer that solve wrote; all zero before the first solve: yes)
__device__ __forceinline__ bool sched_listed(const DevParams& P, int b) {
    if (!P.sched) return true;
    const int32_t* Rd = uniform_ptr(P.sched + (size_t)P.sched_r * P.sched_stride);
    return Rd[kSchedClasses * kSchedCntStride + kSchedClasses * sched_class_len(P.B) + b] >= 0;
}
__device__ __forceinline__ void sched_zero_next(const DevParams& P, int lane) {   // one wave of the launch
    if (P.sched && lane < kSchedClasses) P.sched[(size_t)P.sched_z * P.sched_stride + lane * kSchedCntStride] = 0;
}

}  // namespace brov
