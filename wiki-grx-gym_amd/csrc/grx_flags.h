// grx_flags.h -- the LDS hand-over primitives of the step kernels' wave pipelines (grx_wavepipe.h), in a header of their own so
// that the litmus test tools/micro/lds_handover.hip exercises EXACTLY this code (the protocol is proven, not
// only soaked).  Needs GRX_DEV and FL_COUNT.
//
// -DGRX_SPIN_LIMIT=<polls>: every spin is bounded; a wait that expires stores a code -- 'SPIN', the flag's LDS address, the value
// waited for, the block -- in the host-pinned report word (grx_debug_spin_report) and traps, instead of hanging the GPU inside a
// training job.  The product build spins unbounded (a bound costs the waiter an add and a compare per poll on the critical
// chain); tests/test_env_gpu.py runs every pipelined layout on the bounded build.
#pragma once

#ifdef GRX_SPIN_LIMIT
static __device__ unsigned long long* g_grx_spin_word = nullptr;   // host-pinned (one per translation unit, set by grx_set_spin_word_*)
GRX_DEV void grx_spin_expired(const int* f, int want) {
    if ((threadIdx.x & 63) == 0 && g_grx_spin_word) {
        const unsigned long long code = 0x5350000000000000ull | ((unsigned long long)(blockIdx.x & 0xffffu) << 32) |
                                        ((unsigned long long)((unsigned)(uintptr_t)f & 0xffffu) << 16) | (unsigned long long)(want & 0xffff);
        __hip_atomic_store(g_grx_spin_word, code, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __builtin_trap();
}
#define GRX_SPIN_GUARD(f, want) do { if (++spins_ > (long long)(GRX_SPIN_LIMIT)) grx_spin_expired(f, want); } while (0)
#define GRX_SPIN_DECL long long spins_ = 0
#else
#define GRX_SPIN_GUARD(f, want) do {} while (0)
#define GRX_SPIN_DECL do {} while (0)
#endif

// Hand-over flags and their payload live in LDS: the release / acquire fences are LDS-only ("local" address space), so a
// wave never waits for its global stores or terrain loads in flight when it raises or polls a flag.
GRX_DEV void flag_set(int* f, int v, int lane) {
    // The LDS unit executes a wave's LDS instructions in program order (lgkmcnt returns in order for LDS-only traffic), and the flag is
    // an LDS store like its payload: the payload is in LDS before the flag whatever the fence says.  So the release only has to order
    // the COMPILER's stores (wavefront scope: no s_waitcnt lgkmcnt(0) that would drain the wave's LDS queue, ~100 cycles per hand-over
    // on the producer's chain: +0.9 % on the headline).  -DGRX_FLAG_FENCED restores the workgroup-scope fence.
#ifdef GRX_FLAG_FENCED
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
#endif
    if (lane == 0) __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// a flag that COUNTS its producers (the eight-wave layouts' height scan): same release as flag_set, an LDS atomic add from lane 0
GRX_DEV void flag_add(int* f, int lane) {
#ifdef GRX_FLAG_FENCED
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
#endif
    if (lane == 0) __hip_atomic_fetch_add(f, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
GRX_DEV void flag_wait(int* f, int want) {
    // pure spin (the waiter owns its SIMD; an s_sleep between polls only added detection latency: +1.3 % measured)
    GRX_SPIN_DECL;
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < want) { GRX_SPIN_GUARD(f, want); }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// several flags in ONE poll: lane i < FL_COUNT watches flag i until it reaches want_mine (INT_MIN: not waited for).  A poll is an
// LDS round trip (~100 cycles even when the flag is already up): wave 0 meets ten hand-overs per sub-step
GRX_DEV void flag_wait_all(int* f, int want_mine, int lane) {
    const int* const p = f + (lane < FL_COUNT ? lane : 0);
    GRX_SPIN_DECL;
    while (!__all(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= want_mine)) { GRX_SPIN_GUARD(f, -1); }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
GRX_DEV int flag_want(int lane, int f0, int w0, int f1 = -1, int w1 = 0, int f2 = -1, int w2 = 0, int f3 = -1, int w3 = 0, int f4 = -1, int w4 = 0, int f5 = -1, int w5 = 0,
                      int f6 = -1, int w6 = 0) {
    int w = INT_MIN;
    if (lane == f0) w = w0;
    if (lane == f1) w = w1;
    if (lane == f2) w = w2;
    if (lane == f3) w = w3;
    if (lane == f4) w = w4;
    if (lane == f5) w = w5;
    if (lane == f6) w = w6;
    return w;
}
// block barrier that orders LDS only (global stores stay in flight across it)
GRX_DEV void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

