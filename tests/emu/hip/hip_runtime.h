// TEST INFRASTRUCTURE: a minimal CPU stand-in for <hip/hip_runtime.h> so that sources
// emitted by drstencil can be executed on the host (no GPU in the authoring container).
// Every GPU thread of a workgroup is a ucontext fiber; __syncthreads() yields to a
// round-robin scheduler, so barrier semantics are exact and runs are deterministic.
// Workgroups run one after another.  Used only by tests/test_emulated_kernels.py.
#pragma once
#include <ucontext.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_uint3 { unsigned x, y, z; };
inline emu_uint3 threadIdx, blockIdx;
inline dim3 blockDim, gridDim;

typedef void *hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
inline hipError_t hipGetLastError() { return 0; }
using std::max;
using std::min;

namespace emu {
struct Fiber {
    ucontext_t ctx;
    char *stack = nullptr;
    bool done = false;
    emu_uint3 tid;
};
constexpr size_t kStack = 128 * 1024;
inline std::vector<char *> stack_pool;   // reused across workgroups (no zero fill)
inline ucontext_t sched_ctx;
inline Fiber *cur = nullptr;
inline std::function<void()> *body = nullptr;
inline int xchg[1024];
// a counting barrier (round 3: wave-specialised kernels run loader wavefronts on another code path than the consumers, so "one yield
// per barrier" no longer pairs the right program points -- the consumers' DPP emulation yields too): a fiber leaves __syncthreads()
// only when every live fiber of the workgroup has arrived; fibers that have returned are not waited for (s_barrier semantics)
inline unsigned bar_arrived = 0, bar_gen = 0, live = 0;
inline void release_if_complete() { if (live > 0 && bar_arrived == live) { bar_arrived = 0; bar_gen++; } }

inline void trampoline() {
    (*body)();
    cur->done = true;
    live--;
    release_if_complete();
    swapcontext(&cur->ctx, &sched_ctx);
}
inline void yield() { swapcontext(&cur->ctx, &sched_ctx); }

inline void run_block(dim3 block, std::function<void()> fn) {
    const unsigned nt = block.x * block.y * block.z;
    static std::vector<Fiber> fibers;
    fibers.assign(nt, Fiber());
    while (stack_pool.size() < nt) stack_pool.push_back((char *)malloc(kStack));
    body = &fn;
    bar_arrived = 0; bar_gen = 0; live = nt;
    for (unsigned t = 0; t < nt; t++) {
        Fiber &f = fibers[t];
        f.stack = stack_pool[t];
        f.tid.x = t % block.x; f.tid.y = (t / block.x) % block.y; f.tid.z = t / (block.x * block.y);
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = &sched_ctx;
        makecontext(&f.ctx, (void (*)())trampoline, 0);
    }
    // EMU_ORDER=reverse runs the fibers of every phase last-to-first: a missing barrier between a
    // writer and a reader shows up under at least one of the two orders
    static const bool reverse = getenv("EMU_ORDER") && strcmp(getenv("EMU_ORDER"), "reverse") == 0;
    for (;;) {
        bool any = false;
        for (unsigned i = 0; i < nt; i++) {
            const unsigned t = reverse ? nt - 1 - i : i;
            Fiber &f = fibers[t];
            if (f.done) continue;
            any = true;
            cur = &f;
            threadIdx = f.tid;
            swapcontext(&sched_ctx, &f.ctx);
        }
        if (!any) break;
    }
}

inline void launch(dim3 grid, dim3 block, std::function<void()> fn) {
    gridDim = grid; blockDim = block;
    for (unsigned z = 0; z < grid.z; z++)
        for (unsigned y = 0; y < grid.y; y++)
            for (unsigned x = 0; x < grid.x; x++) {
                blockIdx.x = x; blockIdx.y = y; blockIdx.z = z;
                run_block(block, fn);
            }
}
}  // namespace emu

inline void __syncthreads() {
    const unsigned gen = emu::bar_gen;
    emu::bar_arrived++;
    emu::release_if_complete();
    do emu::yield(); while (emu::bar_gen == gen);
}

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    emu::launch((grid), (block), [&]() { kern(__VA_ARGS__); })

#ifdef DRS_EMULATE
// DPP wave shifts (wave_shr:1 / wave_shl:1, old = 0, bound_ctrl off) across a 64-lane wave
inline int drs_shr_i(int v) {
    const unsigned t = threadIdx.x;
    emu::xchg[t] = v;
    emu::yield();
    int r = (t % 64 == 0) ? 0 : emu::xchg[t - 1];
    emu::yield();
    return r;
}
inline int drs_shl_i(int v) {
    const unsigned t = threadIdx.x;
    const unsigned nt = blockDim.x * blockDim.y * blockDim.z;
    emu::xchg[t] = v;
    emu::yield();
    int r = (t % 64 == 63 || t + 1 >= nt) ? 0 : emu::xchg[t + 1];
    emu::yield();
    return r;
}
#endif
