// sim_rt.cpp -- scheduler of the wave64 functional simulator (see sim_rt.h; TEST INFRASTRUCTURE).
//
// One OS thread runs one workgroup at a time; its threads are fibers (a hand-written x86-64 stack switch).  The scheduler always resumes
// the runnable lane of the FIRST wave in priority order, so a wave runs until it meets a rendezvous nobody else has reached yet: waves
// are maximally skewed against each other, which is what exposes a missing barrier or wait.  A rendezvous of a wave (MFMA, DPP, ...)
// fires when every live lane of the wave has arrived; when nothing can run any more, the lowest pending rendezvous fires with the lanes
// that did arrive -- the others are "inactive lanes" of that instruction, i.e. divergent control flow behaves like EXEC masking.
#include "sim_rt.h"

#include <sys/mman.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

thread_local SimLane* sim_lane = nullptr;
thread_local SimBlockCtx* sim_blk = nullptr;

extern "C" void sim_switch(void** save_sp, void* new_sp);
asm(R"(
.text
.globl sim_switch
.type sim_switch,@function
sim_switch:
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
.size sim_switch,.-sim_switch
)");

namespace {

constexpr size_t STACK_BYTES = 512 * 1024;
constexpr int MAX_LDS = 160 * 1024;

struct DmaEntry { char* dst; int size; unsigned char data[16]; };

enum { ST_READY = 0, ST_WAIT_WAVE, ST_WAIT_RETRY, ST_WAIT_BLOCK, ST_DONE };

struct Fiber {
    SimLane L;
    void* sp = nullptr;
    char* stack = nullptr;
    int state = ST_DONE;
    std::vector<DmaEntry> dma;   // FIFO: [dma_head, size)
    size_t dma_head = 0;
    const void* op_in = nullptr;
    void* op_out = nullptr;
};

struct Wave {
    int first = 0, n = 0, nlive = 0, arrived = 0, op = 0;
    uint64_t imm = 0, arrived_mask = 0, ready_mask = 0, retry_mask = 0;
};

struct BlockRun {
    std::vector<Fiber> fibers;
    std::vector<Wave> waves;
    int nthreads = 0, nlive = 0, bar_arrived = 0;
    void* main_sp = nullptr;
    Fiber* cur = nullptr;
    const std::function<void()>* body = nullptr;
    SimBlockCtx ctx;
    char* lds = nullptr;
    bool descending = false;
};

thread_local BlockRun* g_run = nullptr;

[[noreturn]] void die(const char* msg) {
    fprintf(stderr, "svdx-sim: %s\n", msg);
    abort();
}

void land(Fiber* f, size_t keep) {
    while (f->dma.size() - f->dma_head > keep) {
        const DmaEntry& e = f->dma[f->dma_head++];
        memcpy(e.dst, e.data, e.size);
    }
    if (f->dma_head == f->dma.size()) { f->dma.clear(); f->dma_head = 0; }
}

void yield_to_main() {
    BlockRun* r = g_run;
    Fiber* f = r->cur;
    sim_switch(&f->sp, r->main_sp);
}

inline void set_ready(BlockRun* r, Fiber* f) {
    f->state = ST_READY;
    r->waves[f->L.wave].ready_mask |= 1ull << f->L.lane;
}

float mfma_dot(const SimMfmaIn* a, const SimMfmaIn* b, int kn) {
    // products of 16-bit inputs are exact in fp32; the hardware's accumulation order is not documented -- a double accumulator keeps the
    // model order-independent (the GPU tests hold the kernels to a tolerance, not to bits)
    double s = 0;
    for (int k = 0; k < kn; ++k) s += (double)a->a[k] * (double)b->b[k];
    return (float)s;
}

void fire(BlockRun* r, Wave& w) {
    const uint64_t mask = w.arrived_mask;
    Fiber* fb = &r->fibers[w.first];
    auto in = [&](int l) -> const void* { return (mask >> l & 1) ? fb[l].op_in : nullptr; };
    switch (w.op) {
    case SIM_OP_MFMA32_F16: case SIM_OP_MFMA32_BF16: case SIM_OP_MFMA16_F16: case SIM_OP_MFMA16_BF16: {
        const int kn = (w.op == SIM_OP_MFMA32_F16 || w.op == SIM_OP_MFMA32_BF16) ? 8 : 4;
        // D[i][j] = C[i][j] + sum_k A[i][k] B[k][j]; A: lane l holds row l % 16, k = kn * (l / 16) + 0..kn-1; B: lane l holds column l % 16,
        // same k; C / D: lane l holds column l % 16, rows 4 * (l / 16) + 0..3.  Lanes that are not there contribute zeros.
        for (int l = 0; l < w.n; ++l) {
            if (!(mask >> l & 1)) continue;
            const SimMfmaIn* me = (const SimMfmaIn*)fb[l].op_in;
            float* d = (float*)fb[l].op_out;
            const int col = l & 15;
            for (int e = 0; e < 4; ++e) {
                const int row = 4 * (l >> 4) + e;
                double s = me->c[e];
                for (int g = 0; g < 4; ++g) {
                    const SimMfmaIn* ra = (const SimMfmaIn*)in(g * 16 + row);
                    const SimMfmaIn* cb = (const SimMfmaIn*)in(g * 16 + col);
                    if (ra && cb) s += mfma_dot(ra, cb, kn);
                }
                d[e] = (float)s;
            }
        }
        break;
    }
    case SIM_OP_DPP: {
        const int ctrl = (int)(w.imm & 0xffff), row_mask = (int)(w.imm >> 16 & 0xf), bank_mask = (int)(w.imm >> 20 & 0xf);
        const bool bound = w.imm >> 24 & 1;
        for (int l = 0; l < w.n; ++l) {
            if (!(mask >> l & 1)) continue;
            const int* me = (const int*)fb[l].op_in;
            int src = -1;
            const int row0 = l & ~15, li = l & 15;
            if (ctrl >= 0x00 && ctrl <= 0xff) src = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);
            else if (ctrl >= 0x101 && ctrl <= 0x10f) { const int s = li + (ctrl & 0xf); src = s < 16 ? row0 + s : -1; }          // row_shl
            else if (ctrl >= 0x111 && ctrl <= 0x11f) { const int s = li - (ctrl & 0xf); src = s >= 0 ? row0 + s : -1; }          // row_shr
            else if (ctrl >= 0x121 && ctrl <= 0x12f) src = row0 + ((li - (ctrl & 0xf)) & 15);                                      // row_ror
            else if (ctrl == 0x140) src = row0 + 15 - li;                                                                           // row_mirror
            else if (ctrl == 0x141) src = (l & ~7) | (7 - (l & 7));                                                                 // row_half_mirror
            else die("DPP control not modelled");
            int res;
            const bool enabled = (row_mask >> (l >> 4) & 1) && (bank_mask >> ((l >> 2) & 3) & 1);
            if (!enabled) res = me[0];
            else if (src < 0 || src >= w.n || !(mask >> src & 1)) res = bound ? 0 : me[0];
            else res = ((const int*)fb[src].op_in)[1];
            *(int*)fb[l].op_out = res;
        }
        break;
    }
    case SIM_OP_SHFL_XOR: {
        const int m = (int)(w.imm & 0xffffffffu);
        for (int l = 0; l < w.n; ++l) {
            if (!(mask >> l & 1)) continue;
            const int s = l ^ m;
            const void* sv = (s < w.n) ? in(s) : nullptr;
            memcpy(fb[l].op_out, sv ? sv : fb[l].op_in, 8);
        }
        break;
    }
    case SIM_OP_BPERMUTE: {
        for (int l = 0; l < w.n; ++l) {
            if (!(mask >> l & 1)) continue;
            const int s = (((const int*)fb[l].op_in)[0] >> 2) & 63;
            const int* sv = (const int*)((s < w.n) ? in(s) : nullptr);
            *(int*)fb[l].op_out = sv ? sv[1] : 0;
        }
        break;
    }
    case SIM_OP_READFIRST: {
        const int first = __builtin_ctzll(mask);
        const int v = *(const int*)fb[first].op_in;
        for (int l = 0; l < w.n; ++l) if (mask >> l & 1) *(int*)fb[l].op_out = v;
        break;
    }
    case SIM_OP_TR16: {
        // ds_read_b64_tr_b16: inside each group of 16 lanes, lane i supplies the address of 4 consecutive 16-bit elements (one quarter of a
        // row of a [4][16] block); lane i receives column i of the block: element j comes from lane 4 j + i / 4, its element i % 4.  An
        // address that is not 8-byte aligned reads from the aligned address below it (cdna_hip_programming.md, G17).
        short vals[64][4];
        for (int l = 0; l < w.n; ++l) {
            const void* const* pp = (const void* const*)in(l);
            if (pp) memcpy(vals[l], (const void*)((uintptr_t)*pp & ~(uintptr_t)7), 8);
            else memset(vals[l], 0, 8);
        }
        for (int l = 0; l < w.n; ++l) {
            if (!(mask >> l & 1)) continue;
            short* o = (short*)fb[l].op_out;
            const int g0 = l & ~15, i = l & 15;
            for (int j = 0; j < 4; ++j) o[j] = vals[g0 + 4 * j + (i >> 2)][i & 3];
        }
        break;
    }
    case SIM_OP_ANY: case SIM_OP_ALL: {
        bool any = false, all = true;
        for (int l = 0; l < w.n; ++l) {
            if (!(mask >> l & 1)) continue;
            const bool p = *(const int*)fb[l].op_in != 0;
            any |= p;
            all &= p;
        }
        for (int l = 0; l < w.n; ++l) if (mask >> l & 1) *(int*)fb[l].op_out = w.op == SIM_OP_ANY ? any : all;
        break;
    }
    case SIM_OP_BARRIER: break;
    default: die("unknown wave op");
    }
    for (int l = 0; l < w.n; ++l)
        if (mask >> l & 1) set_ready(r, &fb[l]);
    for (int l = 0; l < w.n; ++l)
        if (w.retry_mask >> l & 1) set_ready(r, &fb[l]);
    w.arrived = 0;
    w.arrived_mask = 0;
    w.retry_mask = 0;
}

void release_barrier(BlockRun* r) {
    for (Fiber& f : r->fibers)
        if (f.state == ST_WAIT_BLOCK) set_ready(r, &f);
    r->bar_arrived = 0;
}

void fiber_entry() {
    BlockRun* r = g_run;
    (*r->body)();
    r = g_run;
    Fiber* f = r->cur;
    land(f, 0);
    f->state = ST_DONE;
    Wave& w = r->waves[f->L.wave];
    --w.nlive;
    --r->nlive;
    if (w.arrived > 0 && w.arrived == w.nlive) fire(r, w);
    if (r->bar_arrived > 0 && r->bar_arrived == r->nlive) release_barrier(r);
    sim_switch(&f->sp, r->main_sp);
    die("resumed a finished fiber");
}

void run_block(BlockRun* r, const std::function<void()>& body, dim3 grid, dim3 block, dim3 bidx, size_t shmem) {
    const int nt = (int)(block.x * block.y * block.z);
    if ((int)r->fibers.size() < nt) {
        r->fibers.resize(nt);                    // (no fiber is live between workgroups, so moving the records is safe)
        for (Fiber& f : r->fibers)
            if (!f.stack) {
                f.stack = (char*)mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
                if (f.stack == MAP_FAILED) die("mmap of a fiber stack failed");
            }
    }
    if (!r->lds) r->lds = (char*)aligned_alloc(256, MAX_LDS + 4096);
    if (shmem > (size_t)MAX_LDS) die("dynamic LDS request beyond 160 KiB");
    memset(r->lds, 0xff, shmem + 256);                       // LDS holds garbage at workgroup start: NaN patterns here
    r->ctx.bidx = bidx;
    r->ctx.bdim = block;
    r->ctx.gdim = grid;
    r->ctx.dyn_lds = r->lds;
    r->body = &body;
    r->nthreads = r->nlive = nt;
    r->bar_arrived = 0;
    const int nw = (nt + 63) / 64;
    r->waves.assign(nw, Wave());
    for (int t = 0; t < nt; ++t) {
        Fiber& f = r->fibers[t];
        f.L.tid = t;
        f.L.tidx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
        f.L.lane = t & 63;
        f.L.wave = t >> 6;
        f.dma.clear();
        f.dma_head = 0;
        void** top = (void**)(f.stack + STACK_BYTES);
        top[-1] = nullptr;                      // fake return address of fiber_entry (never used)
        top[-2] = (void*)&fiber_entry;          // `ret` of the first switch
        for (int i = 3; i <= 8; ++i) top[-i] = nullptr;
        f.sp = (void*)(top - 8);
        Wave& w = r->waves[f.L.wave];
        if (w.n == 0) w.first = t;
        ++w.n;
        ++w.nlive;
        set_ready(r, &f);
    }
    sim_blk = &r->ctx;
    g_run = r;
    while (r->nlive > 0) {
        Fiber* f = nullptr;
        for (int i = 0; i < nw && !f; ++i) {
            Wave& w = r->waves[r->descending ? nw - 1 - i : i];
            if (w.ready_mask) {
                const int l = __builtin_ctzll(w.ready_mask);
                w.ready_mask &= ~(1ull << l);
                f = &r->fibers[w.first + l];
            }
        }
        if (!f) {                                // nothing runnable: the lowest pending rendezvous goes ahead with the lanes it has
            bool fired = false;
            for (int i = 0; i < nw && !fired; ++i) {
                Wave& w = r->waves[r->descending ? nw - 1 - i : i];
                if (w.arrived > 0) { fire(r, w); fired = true; }
            }
            if (!fired) die("deadlock: threads wait at a workgroup barrier that the rest of the workgroup never reaches");
            continue;
        }
        r->cur = f;
        sim_lane = &f->L;
        sim_switch(&r->main_sp, f->sp);
    }
    sim_lane = nullptr;
}

// ---- worker pool: workgroups of one launch are spread over OS threads ----------------------------------------------------------------
struct Pool {
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    const std::function<void()>* body = nullptr;
    dim3 grid, block;
    size_t shmem = 0;
    std::atomic<long> next{0};
    long total = 0;
    int active = 0;
    uint64_t epoch = 0;
    bool descending = false;

    void worker() {
        BlockRun run;
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return epoch != seen; });
                seen = epoch;
            }
            run.descending = descending;
            for (;;) {
                const long b = next.fetch_add(1);
                if (b >= total) break;
                const dim3 bidx((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((long)grid.x * grid.y)));
                run_block(&run, *body, grid, block, bidx, shmem);
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--active == 0) cv_done.notify_all();
            }
        }
    }
};

Pool* pool() {
    static Pool* p = [] {
        Pool* q = new Pool();
        int n = (int)std::thread::hardware_concurrency();
        if (const char* e = getenv("SVDX_SIM_THREADS")) n = atoi(e);
        if (n < 1) n = 1;
        for (int i = 0; i < n; ++i) q->threads.emplace_back([q] { q->worker(); });
        for (auto& t : q->threads) t.detach();
        return q;
    }();
    return p;
}

std::mutex g_launch_mu;

}  // namespace

void sim_launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
    std::lock_guard<std::mutex> launch_lock(g_launch_mu);     // launches are stream-ordered: one at a time
    Pool* p = pool();
    const char* ord = getenv("SVDX_SIM_ORDER");
    {
        std::unique_lock<std::mutex> lk(p->mu);
        p->body = &body;
        p->grid = grid;
        p->block = block;
        p->shmem = shmem;
        p->total = (long)grid.x * grid.y * grid.z;
        p->next = 0;
        p->active = (int)p->threads.size();
        p->descending = ord && ord[0] == 'd';
        ++p->epoch;
    }
    p->cv_work.notify_all();
    std::unique_lock<std::mutex> lk(p->mu);
    p->cv_done.wait(lk, [&] { return p->active == 0; });
}

void sim_wave_op(int op, const void* in, void* out, uint64_t imm) {
    BlockRun* r = g_run;
    Fiber* f = r->cur;
    Wave& w = r->waves[f->L.wave];
    while (w.arrived > 0 && (w.op != op || w.imm != imm)) {    // the wave is split over two different instructions: wait for the other part
        f->state = ST_WAIT_RETRY;
        w.retry_mask |= 1ull << f->L.lane;
        yield_to_main();
    }
    if (w.arrived == 0) { w.op = op; w.imm = imm; }
    f->op_in = in;
    f->op_out = out;
    w.arrived_mask |= 1ull << f->L.lane;
    ++w.arrived;
    if (w.arrived == w.nlive) {
        fire(r, w);
        w.ready_mask &= ~(1ull << f->L.lane);                   // this lane simply keeps running
        f->state = ST_READY;
        return;
    }
    f->state = ST_WAIT_WAVE;
    yield_to_main();
}

void sim_block_barrier(bool fence_vm) {
    BlockRun* r = g_run;
    Fiber* f = r->cur;
    if (fence_vm) land(f, 0);
    ++r->bar_arrived;
    if (r->bar_arrived == r->nlive) {
        f->state = ST_WAIT_BLOCK;
        release_barrier(r);
        r->waves[f->L.wave].ready_mask &= ~(1ull << f->L.lane);
        f->state = ST_READY;
        return;
    }
    f->state = ST_WAIT_BLOCK;
    yield_to_main();
}

void sim_dma(char* lds_dst, const void* src, int size) {
    if (size > 16 || size <= 0) die("LDS-DMA piece larger than 16 bytes");
    // SVDX_SIM_DMA=eager: the other legal extreme -- the piece lands the moment it is issued, so a stage that is refilled while a slower
    // wave still reads it (a write-after-read race the late landing cannot show) gives wrong data
    static const bool eager = getenv("SVDX_SIM_DMA") && getenv("SVDX_SIM_DMA")[0] == 'e';
    if (eager) {
        if (src) memcpy(lds_dst, src, size);
        else memset(lds_dst, 0, size);
        return;
    }
    Fiber* f = g_run->cur;
    DmaEntry e;
    e.dst = lds_dst;
    e.size = size;
    if (src) memcpy(e.data, src, size);
    else memset(e.data, 0, size);
    f->dma.push_back(e);
}

void sim_waitcnt_vm(int n) { land(g_run->cur, (size_t)n); }

const char* sim_buffer_addr(const SimRsrc& rs, int voffset, int soffset, int imm, int size) {
    const uint64_t checked = (uint64_t)(uint32_t)voffset + (uint32_t)imm;
    const uint64_t full = checked + (uint32_t)soffset;
    const bool oob_checked = checked + size > rs.num_records;
    const bool oob_full = full + size > rs.num_records;
    if (oob_checked != oob_full) {
        // the scalar offset decides whether the access is in range: the two readings of the raw-buffer rule disagree here
        static std::atomic<int> warned{0};
        if (warned.fetch_add(1) < 5) fprintf(stderr, "svdx-sim: buffer access whose range check depends on the scalar offset (voffset %d soffset %d records %u)\n", voffset, soffset, rs.num_records);
    }
    if (oob_checked || oob_full) return nullptr;
    return rs.base + full;
}

void sim_buffer_store(const void* v, int size, const SimRsrc& rs, int voffset, int soffset) {
    const char* a = sim_buffer_addr(rs, voffset, soffset, 0, size);
    if (a) memcpy(const_cast<char*>(a), v, size);
}

template <typename T> static T atomic_add_cas(T* p, T v) {
    T old, nw;
    do {
        __atomic_load(p, &old, __ATOMIC_RELAXED);
        nw = old + v;
    } while (!__atomic_compare_exchange(p, &old, &nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    return old;
}
float atomicAdd(float* p, float v) { return atomic_add_cas(p, v); }
double atomicAdd(double* p, double v) { return atomic_add_cas(p, v); }
int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
