// Execution engine of the CUDA-on-the-host shim (see cuda_runtime_api.h).  TEST INFRASTRUCTURE ONLY.
//
// mode 0: a launch is four nested loops (blocks, then threads in x-fastest order); the kernels run this way are
//         the ones whose threads never communicate (integrate, ray-cast, compute_dists, the image kernels).
// mode 1: the threads of one block are ucontext fibers on the calling OS thread.  A fiber runs until it reaches a
//         synchronisation point -- __syncthreads, __ballot/__all, or a cuda_shim::sync_warp() the Makefile's
//         lock-step patch put between the read and the write phase of a warp-synchronous statement -- and the
//         scheduler resumes the fibers in thread order, releasing a barrier when every live thread of the block
//         (or every live lane of the warp) has arrived.  Blocks run one after the other, so the global atomics of
//         the extraction kernel need no locking and its output order is deterministic (block-major).
#include "cuda_runtime_api.h"
#include <ucontext.h>
#include <vector>
#include <cstdio>

namespace cuda_shim
{
thread_local uint3 tl_threadIdx, tl_blockIdx;
thread_local dim3 tl_blockDim, tl_gridDim;
int g_fiber_mode = 0;
int g_parallel_blocks = 0;

namespace
{
enum State { READY, WAIT_BLOCK, WAIT_WARP, DONE };
struct Fiber
{
    ucontext_t ctx;
    char* stack;
    State state;
    unsigned wait_gen;
    uint3 tid;
    int warp;
};
struct BlockRun
{
    std::vector<Fiber> fibers;
    ucontext_t main_ctx;
    const std::function<void()>* body;
    int current;
    int alive;
    int block_arrived;
    unsigned block_gen;
    std::vector<int> warp_alive, warp_arrived;
    std::vector<unsigned> warp_gen, warp_pending, warp_result;
};
thread_local BlockRun* tl_run = 0;
const size_t kStack = 256 * 1024;

void fiber_entry()
{
    BlockRun* r = tl_run;
    (*r->body)();
    Fiber& f = r->fibers[r->current];
    f.state = DONE;
    swapcontext(&f.ctx, &r->main_ctx);
}

void yield_current()
{
    BlockRun* r = tl_run;
    Fiber& f = r->fibers[r->current];
    swapcontext(&f.ctx, &r->main_ctx);
    tl_threadIdx = f.tid;       // (the scheduler also sets it; kept here for clarity)
}

void release_if_complete(BlockRun* r, int warp)
{
    if (r->alive > 0 && r->block_arrived == r->alive) { r->block_arrived = 0; r->block_gen++; }
    if (warp >= 0 && r->warp_alive[warp] > 0 && r->warp_arrived[warp] == r->warp_alive[warp])
    {
        r->warp_arrived[warp] = 0; r->warp_result[warp] = r->warp_pending[warp]; r->warp_pending[warp] = 0; r->warp_gen[warp]++;
    }
}

void run_block_fibers(BlockRun& r, const dim3& block)
{
    const int n = (int)(block.x * block.y * block.z);
    const int nwarps = (n + 31) / 32;
    r.alive = n; r.block_arrived = 0; r.block_gen = 0;
    r.warp_alive.assign(nwarps, 0); r.warp_arrived.assign(nwarps, 0);
    r.warp_gen.assign(nwarps, 0); r.warp_pending.assign(nwarps, 0); r.warp_result.assign(nwarps, 0);
    if ((int)r.fibers.size() < n)
    {
        size_t old = r.fibers.size();
        r.fibers.resize(n);
        for (size_t i = old; i < (size_t)n; ++i) r.fibers[i].stack = (char*)malloc(kStack);
    }
    int t = 0;
    for (unsigned z = 0; z < block.z; ++z) for (unsigned y = 0; y < block.y; ++y) for (unsigned x = 0; x < block.x; ++x, ++t)
    {
        Fiber& f = r.fibers[t];
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = kStack; f.ctx.uc_link = &r.main_ctx;
        makecontext(&f.ctx, fiber_entry, 0);
        f.state = READY; f.tid.x = x; f.tid.y = y; f.tid.z = z; f.warp = t / 32;
        r.warp_alive[f.warp]++;
    }
    while (r.alive > 0)
    {
        bool progressed = false;
        for (int i = 0; i < n; ++i)
        {
            Fiber& f = r.fibers[i];
            bool go = f.state == READY || (f.state == WAIT_BLOCK && f.wait_gen != r.block_gen) ||
                      (f.state == WAIT_WARP && f.wait_gen != r.warp_gen[f.warp]);
            if (!go) continue;
            progressed = true;
            f.state = READY; r.current = i; tl_threadIdx = f.tid;
            swapcontext(&r.main_ctx, &f.ctx);
            if (f.state == DONE)
            {
                r.alive--; r.warp_alive[f.warp]--;
                release_if_complete(&r, f.warp);
            }
        }
        if (!progressed) { fprintf(stderr, "cuda_shim: deadlock (divergent barrier)\n"); abort(); }
    }
}
}  // namespace

void sync_threads()
{
    BlockRun* r = tl_run;
    if (!r) return;                       // mode 0: single thread at a time, a barrier there would be a shim misuse
    Fiber& f = r->fibers[r->current];
    if (++r->block_arrived == r->alive) { r->block_arrived = 0; r->block_gen++; return; }
    f.wait_gen = r->block_gen; f.state = WAIT_BLOCK;
    yield_current();
}

static unsigned warp_arrive(int pred_bit_set, unsigned lane)
{
    BlockRun* r = tl_run;
    Fiber& f = r->fibers[r->current];
    const int w = f.warp;
    if (pred_bit_set) r->warp_pending[w] |= 1u << lane;
    if (++r->warp_arrived[w] == r->warp_alive[w])
    {
        r->warp_arrived[w] = 0; r->warp_result[w] = r->warp_pending[w]; r->warp_pending[w] = 0; r->warp_gen[w]++;
        return r->warp_result[w];
    }
    f.wait_gen = r->warp_gen[w]; f.state = WAIT_WARP;
    yield_current();
    return r->warp_result[w];
}

unsigned int lane_id()
{
    return (tl_threadIdx.z * tl_blockDim.x * tl_blockDim.y + tl_threadIdx.y * tl_blockDim.x + tl_threadIdx.x) & 31u;
}

void sync_warp()
{
    if (!tl_run) { fprintf(stderr, "cuda_shim: warp-synchronous code needs fiber mode\n"); abort(); }
    warp_arrive(0, lane_id());
}

unsigned int warp_ballot(int pred)
{
    if (!tl_run) { fprintf(stderr, "cuda_shim: __ballot/__all need fiber mode\n"); abort(); }
    return warp_arrive(pred != 0, lane_id());
}

void launch(const LaunchCfg& cfg, const std::function<void()>& body)
{
    const dim3 grid = cfg.grid, block = cfg.block;
    if (g_fiber_mode)
    {
        static thread_local BlockRun run;
        run.body = &body;
        tl_run = &run;
        tl_gridDim = grid; tl_blockDim = block;
        for (unsigned bz = 0; bz < grid.z; ++bz) for (unsigned by = 0; by < grid.y; ++by) for (unsigned bx = 0; bx < grid.x; ++bx)
        {
            tl_blockIdx.x = bx; tl_blockIdx.y = by; tl_blockIdx.z = bz;
            run_block_fibers(run, block);
        }
        tl_run = 0;
        return;
    }
    const long nblocks = (long)grid.x * grid.y * grid.z;
    const int par = g_parallel_blocks;
#pragma omp parallel for schedule(dynamic, 4) if (par)
    for (long b = 0; b < nblocks; ++b)
    {
        tl_gridDim = grid; tl_blockDim = block;
        tl_blockIdx.x = (unsigned)(b % grid.x); tl_blockIdx.y = (unsigned)((b / grid.x) % grid.y); tl_blockIdx.z = (unsigned)(b / ((long)grid.x * grid.y));
        for (unsigned z = 0; z < block.z; ++z) for (unsigned y = 0; y < block.y; ++y) for (unsigned x = 0; x < block.x; ++x)
        {
            tl_threadIdx.x = x; tl_threadIdx.y = y; tl_threadIdx.z = z;
            body();
        }
    }
}
}  // namespace cuda_shim
