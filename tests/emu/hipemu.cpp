// TEST-ONLY runtime for tests/emu/shim/hip/hip_runtime.h: runs one workgroup at a time on
// cooperative fibers (one fiber per work-item), several workgroups in parallel on OS threads.
// Fibers of a workgroup are scheduled round-robin and only switch at __syncthreads() or at a
// wave64 collective, so execution between sync points is sequential per lane: a missing barrier
// between "lane A writes" and "lane B reads" shows up deterministically (lane order 0..N-1, or
// reversed with HIPEMU_REVERSE=1 to expose write-after-read hazards).
#include "hip/hip_runtime.h"

#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hipemu_switch, .-hipemu_switch
)");

// dynamic LDS region of the kernels (`extern __shared__ unsigned char dtqn_smem[]`)
alignas(64) thread_local unsigned char dtqn_smem[163840];

namespace hipemu {

thread_local Ctx* g_ctx = nullptr;
static std::atomic<int> g_err{0};

enum State { RUN, WAVE_WAIT, BLOCK_WAIT, DONE };

struct Fiber {
    Ctx ctx;
    void* sp = nullptr;
    State st = RUN;
    int lane = 0, wave = 0;
    unsigned char* stack = nullptr;
};

struct Wave {
    int waiting = 0;
    int live = 0;
    int phase = 0;
    int words = 0;                     // words per lane of the exchange in flight
    uint32_t buf[2][64 * 4];
};

struct Block {
    std::vector<Fiber> fibers;
    std::vector<Wave> waves;
    int block_waiting = 0;
    int live = 0;
    void* sched_sp = nullptr;
    Fiber* cur = nullptr;
    const std::function<void()>* body = nullptr;
};

static thread_local Block* t_blk = nullptr;
static const size_t kStack = 256 * 1024;

static void yield_to_sched() {
    Block* b = t_blk;
    Fiber* f = b->cur;
    hipemu_switch(&f->sp, b->sched_sp);
}

static void fiber_main() {
    Block* b = t_blk;
    (*b->body)();
    Fiber* f = b->cur;
    f->st = DONE;
    b->live--;
    b->waves[f->wave].live--;
    yield_to_sched();
    std::abort();  // never resumed
}

void block_barrier() {
    Block* b = t_blk;
    Fiber* f = b->cur;
    f->st = BLOCK_WAIT;
    b->block_waiting++;
    yield_to_sched();
}

int lane_id() { return t_blk->cur->lane; }

const uint32_t* wave_exchange(const uint32_t* mine, int n) {
    Block* b = t_blk;
    Fiber* f = b->cur;
    Wave& w = b->waves[f->wave];
    const int ph = w.phase & 1;
    if (n > 4) std::abort();
    if (w.waiting == 0) {
        w.words = n;
        // lanes that are not part of the launch (partial last wave) read as zero
        std::memset(w.buf[ph], 0, sizeof(w.buf[ph]));
    } else if (w.words != n) {
        std::fprintf(stderr, "hipemu: divergent wave collective (wave %d)\n", f->wave);
        std::abort();
    }
    for (int i = 0; i < n; ++i) w.buf[ph][f->lane * n + i] = mine[i];
    f->st = WAVE_WAIT;
    w.waiting++;
    yield_to_sched();
    return w.buf[ph];
}

void atomic_add_f32(float* p, float v) {
    uint32_t* u = reinterpret_cast<uint32_t*>(p);
    uint32_t old = __atomic_load_n(u, __ATOMIC_RELAXED);
    for (;;) {
        float f;
        std::memcpy(&f, &old, 4);
        f += v;
        uint32_t nw;
        std::memcpy(&nw, &f, 4);
        if (__atomic_compare_exchange_n(u, &old, nw, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return;
    }
}

hipError_t last_error() { return g_err.exchange(0); }

static void run_block(Block& blk, dim3 grid, dim3 block, dim3 bid, const std::function<void()>& body, bool reverse) {
    const int nthreads = (int)(block.x * block.y * block.z);
    const int nwaves = (nthreads + 63) / 64;
    blk.waves.assign(nwaves, Wave());
    blk.block_waiting = 0;
    blk.live = nthreads;
    blk.body = &body;
    if ((int)blk.fibers.size() < nthreads) {
        size_t old = blk.fibers.size();
        blk.fibers.resize(nthreads);
        for (size_t i = old; i < blk.fibers.size(); ++i) blk.fibers[i].stack = (unsigned char*)std::aligned_alloc(64, kStack);
    }
    for (int t = 0; t < nthreads; ++t) {
        Fiber& f = blk.fibers[t];
        f.ctx.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
        f.ctx.bid = bid;
        f.ctx.bdim = block;
        f.ctx.gdim = grid;
        f.lane = t & 63;
        f.wave = t >> 6;
        f.st = RUN;
        blk.waves[f.wave].live++;
        // initial frame: 6 callee-saved regs, then return address = fiber_main, then a fake return slot
        uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
        void** sp = (void**)(top - 16 - 48);
        for (int i = 0; i < 6; ++i) sp[i] = nullptr;
        sp[6] = (void*)&fiber_main;
        sp[7] = nullptr;
        f.sp = sp;
    }
    t_blk = &blk;
    int guard = 0;
    while (blk.live > 0) {
        bool progressed = false;
        for (int k = 0; k < nthreads; ++k) {
            const int t = reverse ? nthreads - 1 - k : k;
            Fiber& f = blk.fibers[t];
            if (f.st != RUN) continue;
            blk.cur = &f;
            g_ctx = &f.ctx;
            hipemu_switch(&blk.sched_sp, f.sp);
            progressed = true;
        }
        // release wave collectives that every live lane has reached
        for (int w = 0; w < nwaves; ++w) {
            Wave& wv = blk.waves[w];
            if (wv.waiting > 0 && wv.waiting == wv.live) {
                wv.waiting = 0;
                wv.phase++;
                for (int t = w * 64; t < std::min(nthreads, (w + 1) * 64); ++t)
                    if (blk.fibers[t].st == WAVE_WAIT) blk.fibers[t].st = RUN;
                progressed = true;
            }
        }
        if (blk.block_waiting > 0 && blk.block_waiting == blk.live) {
            blk.block_waiting = 0;
            for (int t = 0; t < nthreads; ++t)
                if (blk.fibers[t].st == BLOCK_WAIT) blk.fibers[t].st = RUN;
            progressed = true;
        }
        if (!progressed && ++guard > 2) {
            std::fprintf(stderr, "hipemu: deadlock in block (%u,%u,%u): %d live, %d at __syncthreads, divergent barrier or collective?\n",
                         bid.x, bid.y, bid.z, blk.live, blk.block_waiting);
            std::abort();
        }
        if (progressed) guard = 0;
    }
    t_blk = nullptr;
    g_ctx = nullptr;
}

// fiber stacks are expensive to allocate: keep finished Blocks in a pool across launches
static std::mutex g_pool_mu;
static std::vector<Block*> g_pool;
static Block* acquire_block() {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (g_pool.empty()) return new Block();
    Block* b = g_pool.back();
    g_pool.pop_back();
    return b;
}
static void release_block(Block* b) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool.push_back(b);
}

void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body) {
    if (smem > sizeof(dtqn_smem) || block.x * block.y * block.z > 1024 || block.x * block.y * block.z == 0) {
        g_err = hipErrorInvalidValue;
        return;
    }
    const long nblocks = (long)grid.x * grid.y * grid.z;
    if (nblocks == 0) return;
    const char* rv = std::getenv("HIPEMU_REVERSE");
    const bool reverse = rv && rv[0] == '1';
    int nthr = (int)std::min<long>(nblocks, std::max(1u, std::thread::hardware_concurrency()));
    const char* nt = std::getenv("HIPEMU_THREADS");
    if (nt) nthr = std::max(1, std::min(nthr, std::atoi(nt)));
    std::atomic<long> next{0};
    auto worker = [&]() {
        Block* pb = acquire_block();
        Block& blk = *pb;
        for (;;) {
            long i = next.fetch_add(1);
            if (i >= nblocks) break;
            dim3 bid((unsigned)(i % grid.x), (unsigned)((i / grid.x) % grid.y), (unsigned)(i / ((long)grid.x * grid.y)));
            run_block(blk, grid, block, bid, body, reverse);
        }
        release_block(pb);
    };
    if (nthr == 1) {
        worker();
    } else {
        std::vector<std::thread> ts;
        for (int i = 0; i < nthr; ++i) ts.emplace_back(worker);
        for (auto& t : ts) t.join();
    }
}

}  // namespace hipemu
