// TEST-ONLY: CPU executor behind tests/emu/include/hip/hip_runtime.h (see the note there).
// One OS worker per concurrently running workgroup; inside a worker every work-item is a ucontext
// fiber, scheduled round-robin and parked at __syncthreads / wave collectives.
#include <hip/hip_runtime.h>
#include <ucontext.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

namespace emu {

enum State { READY, WAIT_BLOCK, WAIT_WAVE, DONE };
constexpr size_t STACK = 256 * 1024;
constexpr int SLOT = 32;

struct Wave {
    int live = 0, arrived = 0;
    alignas(16) unsigned char buf[2][64][SLOT];
};

struct Lane {
    dim3 tid;
    int linear = 0;
    State state = READY;
    unsigned round = 0;
    ucontext_t ctx;
    unsigned char* stack = nullptr;
};

struct Block {
    std::vector<Lane> lanes;
    std::vector<Wave> waves;
    int live = 0, arrived = 0;
    ucontext_t sched;
    void (*tramp)(void*) = nullptr;
    void* args = nullptr;
};

thread_local Lane* cur = nullptr;
thread_local dim3 t_bid, t_bdim, t_gdim;
static thread_local Block* blk = nullptr;
// fiber stacks of the calling thread, released when the (per-launch) worker thread ends
struct StackPool { std::vector<unsigned char*> v; ~StackPool() { for (auto* p : v) std::free(p); } };
static thread_local StackPool stack_pool;

const dim3& cur_tid() { return cur->tid; }
int lane_id() { return cur->linear & 63; }

static void yield_to_sched() { swapcontext(&cur->ctx, &blk->sched); }
void relax() { std::this_thread::yield(); yield_to_sched(); }      // state stays READY: the scheduler comes back round-robin

static void release_block() {
    for (auto& l : blk->lanes)
        if (l.state == WAIT_BLOCK) l.state = READY;
    blk->arrived = 0;
}
static void release_wave(int w) {
    int lo = w * 64, hi = std::min<int>(lo + 64, (int)blk->lanes.size());
    for (int i = lo; i < hi; ++i)
        if (blk->lanes[i].state == WAIT_WAVE) blk->lanes[i].state = READY;
    blk->waves[w].arrived = 0;
}

void sync_block() {
    cur->state = WAIT_BLOCK;
    if (++blk->arrived == blk->live) release_block();
    if (cur->state != READY) yield_to_sched();
}

const unsigned char* wave_exchange(const void* mine, size_t n) {
    if (n > (size_t)SLOT) { std::fprintf(stderr, "emu: exchange too wide\n"); std::abort(); }
    int w = cur->linear >> 6;
    Wave& wv = blk->waves[w];
    unsigned idx = cur->round & 1u;
    std::memcpy(wv.buf[idx][cur->linear & 63], mine, n);
    cur->round++;
    cur->state = WAIT_WAVE;
    if (++wv.arrived == wv.live) release_wave(w);
    if (cur->state != READY) yield_to_sched();
    // callers index the table with stride = their own element size; repack to that stride
    static thread_local unsigned char packed[64 * SLOT];
    for (int i = 0; i < 64; ++i) std::memcpy(packed + (size_t)i * n, wv.buf[idx][i], n);
    return packed;
}

static void lane_entry() {
    blk->tramp(blk->args);
    Lane* me = cur;
    me->state = DONE;
    blk->live--;
    Wave& wv = blk->waves[me->linear >> 6];
    wv.live--;
    if (blk->live > 0 && blk->arrived == blk->live) release_block();
    if (wv.live > 0 && wv.arrived == wv.live) release_wave(me->linear >> 6);
    swapcontext(&me->ctx, &blk->sched);
}

static void run_block(void (*tramp)(void*), void* args, dim3 bid, dim3 grid, dim3 block) {
    Block b;
    blk = &b;
    t_bid = bid; t_bdim = block; t_gdim = grid;
    int nt = (int)(block.x * block.y * block.z);
    b.lanes.resize(nt);
    b.waves.resize((nt + 63) / 64);
    b.live = nt;
    b.tramp = tramp; b.args = args;
    std::vector<unsigned char*>* stacks = &stack_pool.v;
    while ((int)stacks->size() < nt) stacks->push_back(static_cast<unsigned char*>(std::malloc(STACK)));
    for (int i = 0; i < nt; ++i) {
        Lane& l = b.lanes[i];
        l.linear = i;
        l.tid = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
        l.stack = (*stacks)[i];
        b.waves[i >> 6].live++;
        getcontext(&l.ctx);
        l.ctx.uc_stack.ss_sp = l.stack;
        l.ctx.uc_stack.ss_size = STACK;
        l.ctx.uc_link = &b.sched;
        makecontext(&l.ctx, (void (*)())lane_entry, 0);
    }
    int done = 0;
    while (done < nt) {
        bool progress = false;
        for (int i = 0; i < nt; ++i) {
            Lane& l = b.lanes[i];
            if (l.state != READY) continue;
            progress = true;
            cur = &l;
            swapcontext(&b.sched, &l.ctx);
            if (l.state == DONE) done++;
        }
        if (!progress) {
            std::fprintf(stderr, "emu: deadlock in block (%u,%u): %d/%d done, block barrier %d/%d\n", bid.x, bid.y,
                         done, nt, b.arrived, b.live);
            for (size_t wv = 0; wv < b.waves.size(); ++wv) {
                int nw = 0, nb = 0;
                for (int i = (int)wv * 64; i < std::min<int>((int)wv * 64 + 64, nt); ++i) { nw += b.lanes[i].state == WAIT_WAVE; nb += b.lanes[i].state == WAIT_BLOCK; }
                std::fprintf(stderr, "emu:   wave %zu: %d lanes in a wave collective, %d at the block barrier\n", wv, nw, nb);
            }
            std::abort();
        }
    }
    cur = nullptr;
    blk = nullptr;
}

static int n_workers() {
    const char* e = std::getenv("RDMI_EMU_THREADS");
    int n = e ? std::atoi(e) : (int)std::thread::hardware_concurrency();
    return n < 4 ? 4 : (n > 16 ? 16 : n);       // >= 4: the four workgroups of a co-operative group (consecutive ids here) must run concurrently
}

void launch(void (*tramp)(void*), void* args, dim3 grid, dim3 block, size_t lds) {
    if (lds > 160 * 1024) { std::fprintf(stderr, "emu: LDS request %zu > 160 KiB\n", lds); std::abort(); }
    size_t nb = (size_t)grid.x * grid.y * grid.z;
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (;;) {
            size_t i = next.fetch_add(1);
            if (i >= nb) break;
            dim3 bid((unsigned)(i % grid.x), (unsigned)((i / grid.x) % grid.y), (unsigned)(i / ((size_t)grid.x * grid.y)));
            run_block(tramp, args, bid, grid, block);
        }
    };
    int nw = (int)std::min<size_t>(nb, (size_t)n_workers());
    if (nw <= 1) { work(); return; }
    std::vector<std::thread> th;
    for (int i = 0; i < nw; ++i) th.emplace_back(work);
    for (auto& t : th) t.join();
}

}  // namespace emu

// the one dynamic-LDS symbol every kernel declares (extern __shared__ -> extern thread_local)
alignas(16) thread_local unsigned char rdmi_lds[160 * 1024];

struct emuEvent { std::chrono::steady_clock::time_point t; };
hipError_t hipEventCreate(hipEvent_t* e) { *e = new emuEvent(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
