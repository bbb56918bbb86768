// hipemu -- a lane-accurate CPU interpreter for the gfx950 kernels of pytorch-studiogan_amd/csrc, TEST INFRASTRUCTURE ONLY.
//
// The kernel sources are compiled for the HOST (x86, amdclang++) against this header instead of the HIP runtime, after tests/hipemu/translate.py
// has rewritten the constructs a host compiler cannot take (inline gfx950 assembly, address-space casts, `__shared__` declarations). Every
// thread of a workgroup is a fibre (a hand-written stack switch); the cross-lane instructions the kernels are built from -- MFMA, the LDS transpose read,
// LDS-DMA (`buffer_load ... lds`), DPP, shuffles, barriers, `s_waitcnt vmcnt` -- are executed with the semantics MI355X_MICROARCH.md /
// cdna_hip_programming.md document, per wave, once all live lanes of the wave have arrived. What this buys: the index arithmetic, fragment
// layouts, staging addresses and result layouts of a kernel are checked on a machine without a GPU, and LDS-DMA completion can be made
// adversarial (a transfer lands only when a `s_waitcnt vmcnt(n)` / barrier-with-fence forces it, or immediately: HIPEMU_DMA=late|eager) and
// waves can be scheduled greedily in a seeded order (HIPEMU_SCHED=seed), so missing waits and buffers re-used too early show up as wrong
// results. What it does not model: timing, bank conflicts, register pressure, the rounding order inside an MFMA (fp32 left-to-right here).
// The semantics themselves are pinned by running kernels that HAVE passed on the GPU through it (tests/test_hipemu_cpu.py).
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <type_traits>
#include <deque>
#include <functional>
#include <map>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }

__asm__(R"(
.text
.weak hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq (%rsi), %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size hipemu_switch,.-hipemu_switch
)");

namespace hipemu {

typedef __attribute__((ext_vector_type(4))) uint32_t e_u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t e_u32x2;
typedef __attribute__((ext_vector_type(16))) float e_f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 e_bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 e_bf16x2;
typedef __attribute__((ext_vector_type(4))) short e_s16x4;

struct U3 { unsigned x, y, z; };
struct DmaOp { char* dst; int bytes; uint64_t mask; uint8_t data[64][16]; bool nop; };
struct Wave;
struct Wave {
  int nlive = 0, arrived = 0;
  unsigned gen = 0;
  const void* in[64];
  void* out[64];
  uint64_t amask = 0;
  std::function<void(Wave&)> fire;
  std::deque<DmaOp> dma;
};
// fibre switch: callee-saved registers + stack pointer (System V x86-64). glibc's swapcontext makes a signal-mask system call per switch (~0.5 us);
// this is ~10 ns, and a kernel launch is tens of thousands to millions of switches.
struct Ctx { void* rsp; };
extern "C" void hipemu_switch(Ctx* from, Ctx* to);
struct Lane {
  Ctx ctx;
  char* stack = nullptr;
  bool done = false;
  int wave = 0, lane = 0;
  U3 tid;
  int wait_kind = 0;      // 0 runnable, 1 wave collective, 2 workgroup barrier
  unsigned wait_gen = 0;
  // results of inline-assembly LDS reads (ds_read_b64_tr_b16) that have been issued but not waited for: the destination holds poison until a
  // `s_waitcnt lgkmcnt(n)` leaves at most n of them outstanding (LDS operations of a wave return in order)
  struct PendingRead { void* dst; unsigned char val[8]; };
  std::deque<PendingRead> pending;
};
struct Block {
  std::vector<Lane> lanes;
  std::vector<Wave> waves;
  int nlive = 0, bar_arrived = 0;
  unsigned bar_gen = 0;
  char* lds = nullptr;
  size_t dyn_size = 0, static_top = 0;
  std::map<int, size_t> statics;
  U3 bid;
};
struct Counters { long launches = 0, blocks = 0, mfma = 0, dma_ops = 0, tr_reads = 0; };
struct Config : Counters {
  int dma_late = 0;          // 1: an LDS-DMA transfer lands when a wait forces it; 0: when it is issued
  int greedy = 0;            // 1: a wave runs until it blocks at a workgroup barrier before the next wave gets a turn
  unsigned seed = 0;         // wave order permutation
  int threads = 0;           // host threads a launch spreads its workgroups over (0: HIPEMU_THREADS or min(8, cores); 1: sequential, deterministic atomics)
};
inline Config g_cfg;                                  // switches + totals (the per-thread counts are merged in when a launch ends)
inline std::mutex g_cnt_mutex;
// per host thread: the workgroup it is running (a fibre never migrates), its scheduler context, its LDS, its counters
inline thread_local Counters t_cnt;
inline thread_local Block* g_blk = nullptr;
inline thread_local Lane* g_cur = nullptr;
inline thread_local Ctx g_main;
inline thread_local char* g_lds_arena = nullptr;
inline thread_local std::vector<char*>* t_stacks = nullptr;
inline U3 g_grid, g_bdim;                             // per launch (launches do not overlap)
inline const std::function<void()>* g_kernel = nullptr;
constexpr size_t LDS_BYTES = 160 * 1024;
constexpr size_t STACK_BYTES = 192 * 1024;

[[noreturn]] inline void die(const char* msg) {
  fprintf(stderr, "hipemu: %s (block %u,%u thread %u)\n", msg, g_blk ? g_blk->bid.x : 0, g_blk ? g_blk->bid.y : 0, g_cur ? g_cur->tid.x : 0);
  abort();
}
inline void yield_to_main() { hipemu_switch(&g_cur->ctx, &g_main); }

inline void apply_dma(const DmaOp& op) {
  if (op.nop) return;
  for (int l = 0; l < 64; l++)
    if (op.mask >> l & 1) {
      char* d = op.dst + (size_t)l * op.bytes;
      if (d < g_blk->lds || d + op.bytes > g_blk->lds + LDS_BYTES) die("LDS-DMA destination outside the workgroup's LDS");
      memcpy(d, op.data[l], op.bytes);
    }
}
inline void drain_dma(Wave& w, size_t keep) {
  while (w.dma.size() > keep) { apply_dma(w.dma.front()); w.dma.pop_front(); }
}

// a wave-level collective: every live lane of the wave deposits (in, out); the last one to arrive runs `fire` for all of them
template <class F>
inline void wave_sync(const void* in, void* out, F fire) {
  Lane* me = g_cur;
  Wave& w = g_blk->waves[me->wave];
  w.in[me->lane] = in; w.out[me->lane] = out; w.amask |= 1ull << me->lane;
  w.arrived++;
  if (w.arrived == w.nlive) {
    fire(w);
    w.arrived = 0; w.amask = 0; w.gen++; w.fire = nullptr;
  } else {
    if (!w.fire) w.fire = [fire](Wave& ww) { fire(ww); };
    me->wait_kind = 1; me->wait_gen = w.gen;
    yield_to_main();
    me->wait_kind = 0;
  }
}
inline void block_barrier() {
  Lane* me = g_cur;
  Block& b = *g_blk;
  b.bar_arrived++;
  if (b.bar_arrived == b.nlive) { b.bar_arrived = 0; b.bar_gen++; }
  else { me->wait_kind = 2; me->wait_gen = b.bar_gen; yield_to_main(); me->wait_kind = 0; }
}

inline void lane_entry() {
  (*g_kernel)();
  Lane* me = g_cur;
  Block& b = *g_blk;
  Wave& w = b.waves[me->wave];
  me->done = true;
  w.nlive--; b.nlive--;
  if (w.arrived && w.arrived == w.nlive && w.fire) { auto f = w.fire; f(w); w.arrived = 0; w.amask = 0; w.gen++; w.fire = nullptr; }
  if (w.nlive == 0) drain_dma(w, 0);
  if (b.bar_arrived && b.bar_arrived == b.nlive) { b.bar_arrived = 0; b.bar_gen++; }
  hipemu_switch(&me->ctx, &g_main);          // never resumed
  abort();
}
inline void prepare_fibre(Lane& L) {
  // the first switch into the fibre pops six zeros into the callee-saved registers and returns into lane_entry with the stack 8 mod 16
  uintptr_t top = ((uintptr_t)L.stack + STACK_BYTES) & ~(uintptr_t)15;
  void** sp = (void**)(top - 64);
  sp[6] = (void*)&lane_entry;                // return address at an address = 0 mod 16 (sp + 6 words = top - 16)
  for (int i = 0; i < 6; i++) sp[i] = nullptr;
  sp[7] = nullptr;
  L.ctx.rsp = sp;
}

inline void configure_from_env() {
  static bool done = false;
  if (done) return;
  done = true;
  if (const char* m = getenv("HIPEMU_DMA")) g_cfg.dma_late = (m[0] == 'l');
  if (const char* m = getenv("HIPEMU_SCHED")) { g_cfg.greedy = 1; g_cfg.seed = (unsigned)strtoul(m, nullptr, 10); }
  if (const char* m = getenv("HIPEMU_THREADS")) g_cfg.threads = atoi(m);
}

inline void run_block(unsigned lin, dim3 grid, dim3 block, size_t lds) {
  const int nthreads = block.x * block.y * block.z;
  if (!g_lds_arena) {
    g_lds_arena = (char*)mmap(nullptr, LDS_BYTES + 4096, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_32BIT, -1, 0);
    if (g_lds_arena == (char*)MAP_FAILED) { perror("mmap"); abort(); }
    g_lds_arena += 1024;            // LDS address 0 is not where a kernel's image starts in this model: forgetting the base shows
    // LDS is NOT cleared between workgroups: a kernel that reads what it never wrote sees the previous workgroup's bytes (first one: 0xCD)
    memset(g_lds_arena, 0xCD, LDS_BYTES);
  }
  if (!t_stacks) t_stacks = new std::vector<char*>();
  while ((int)t_stacks->size() < nthreads) t_stacks->push_back((char*)malloc(STACK_BYTES));
  const unsigned bx = lin % grid.x, by = (lin / grid.x) % grid.y, bz = lin / (grid.x * grid.y);
  Block b;
  b.bid = {bx, by, bz};
  b.lds = g_lds_arena; b.dyn_size = lds; b.static_top = (lds + 15) & ~(size_t)15;
  b.lanes.resize(nthreads); b.waves.resize((nthreads + 63) / 64);        // (a partial last wave: its missing lanes never exist)
  b.nlive = nthreads;
  g_blk = &b;
  for (int t = 0; t < nthreads; t++) {
    Lane& L = b.lanes[t];
    L.wave = t / 64; L.lane = t % 64;
    L.tid = {(unsigned)(t % block.x), (unsigned)((t / block.x) % block.y), (unsigned)(t / (block.x * block.y))};
    L.stack = (*t_stacks)[t];
    prepare_fibre(L);
    b.waves[L.wave].nlive++;
  }
  std::vector<int> order(b.waves.size());
  for (size_t i = 0; i < order.size(); i++) order[i] = (int)i;
  if (g_cfg.greedy) {
    unsigned s = g_cfg.seed * 2654435761u + lin * 40503u + 12345u;
    for (size_t i = order.size(); i > 1; i--) { s = s * 1664525u + 1013904223u; std::swap(order[i - 1], order[(s >> 16) % i]); }
  }
  int ndone = 0;
  while (ndone < nthreads) {
    bool progress = false;
    for (int wi : order) {
      bool ran;
      do {
        ran = false;
        for (int l = 0; l < 64 && wi * 64 + l < nthreads; l++) {
          Lane& L = b.lanes[wi * 64 + l];
          if (L.done) continue;
          const bool ok = L.wait_kind == 0 || (L.wait_kind == 1 && b.waves[wi].gen != L.wait_gen) || (L.wait_kind == 2 && b.bar_gen != L.wait_gen);
          if (!ok) continue;
          g_cur = &L;
          hipemu_switch(&g_main, &L.ctx);
          if (L.done) ndone++;
          ran = progress = true;
        }
      } while (g_cfg.greedy && ran);
    }
    if (!progress) { g_cur = nullptr; die("deadlock: a collective or barrier that not every live lane reaches"); }
  }
  t_cnt.blocks++;
  g_blk = nullptr; g_cur = nullptr;
}

// persistent worker threads: run(n, job) executes job on n threads (the caller is one of them) and returns when all are done
struct Pool {
  std::mutex m;
  std::condition_variable cv_work, cv_done;
  std::vector<std::thread> workers;
  const std::function<void()>* job = nullptr;
  unsigned gen = 0; int want = 0, started = 0, finished = 0;
  bool stop = false;
  void worker() {
    unsigned seen = 0;
    for (;;) {
      const std::function<void()>* j;
      {
        std::unique_lock<std::mutex> g(m);
        cv_work.wait(g, [&] { return stop || (gen != seen && started < want); });
        if (stop) return;
        seen = gen; started++; j = job;
      }
      (*j)();
      { std::lock_guard<std::mutex> g(m); finished++; }
      cv_done.notify_all();
    }
  }
  void run(int n, const std::function<void()>& f) {
    while ((int)workers.size() < n - 1) workers.emplace_back([this] { worker(); });
    { std::lock_guard<std::mutex> g(m); job = &f; want = n - 1; started = 0; finished = 0; gen++; }
    cv_work.notify_all();
    f();
    std::unique_lock<std::mutex> g(m);
    cv_done.wait(g, [&] { return finished == want; });
    want = 0;
  }
  ~Pool() {
    { std::lock_guard<std::mutex> g(m); stop = true; }
    cv_work.notify_all();
    for (auto& t : workers) t.join();
  }
};

inline void launch(dim3 grid, dim3 block, size_t lds, const std::function<void()>& kernel) {
  configure_from_env();
  const int nthreads = block.x * block.y * block.z;
  if (lds > LDS_BYTES) { fprintf(stderr, "hipemu: %zu B of dynamic LDS\n", lds); abort(); }
  g_grid = {grid.x, grid.y, grid.z}; g_bdim = {block.x, block.y, block.z};
  g_kernel = &kernel;
  const unsigned nblocks = grid.x * grid.y * grid.z;
  int T = g_cfg.threads;
  if (T <= 0) { T = (int)sysconf(_SC_NPROCESSORS_ONLN); if (T > 8) T = 8; if (T < 1) T = 1; }
  if ((unsigned)T > nblocks) T = (int)nblocks;
  auto merge = [] {
    std::lock_guard<std::mutex> g(g_cnt_mutex);
    g_cfg.blocks += t_cnt.blocks; g_cfg.mfma += t_cnt.mfma; g_cfg.dma_ops += t_cnt.dma_ops; g_cfg.tr_reads += t_cnt.tr_reads;
    t_cnt = Counters();
  };
  if (T <= 1) {
    for (unsigned b = 0; b < nblocks; b++) run_block(b, grid, block, lds);
    merge();
  } else {
    // workgroups are independent (the kernels' only inter-workgroup traffic is accumulate-style atomics, real atomics here): spread them over a
    // persistent pool of host threads (each keeps its LDS arena and fibre stacks between launches)
    static Pool pool;
    std::atomic<unsigned> next{0};
    pool.run(T, [&] {
      for (unsigned b = next.fetch_add(1); b < nblocks; b = next.fetch_add(1)) run_block(b, grid, block, lds);
      merge();
    });
  }
  g_cfg.launches++;
}

// ---- LDS ------------------------------------------------------------------------------------------------------------------------------------
inline char* dyn_lds() { return g_blk->lds; }
inline void* static_lds(size_t bytes, int id) {
  Block& b = *g_blk;
  auto it = b.statics.find(id);
  if (it == b.statics.end()) {
    const size_t off = b.static_top;
    b.static_top = (off + bytes + 15) & ~(size_t)15;
    if (b.static_top > LDS_BYTES) die("static LDS overflow");
    it = b.statics.emplace(id, off).first;
  }
  return b.lds + it->second;
}
inline unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)p; }      // the arena sits below 4 GB (MAP_32BIT)
inline char* lds_ptr(unsigned a) {
  char* p = (char*)(uintptr_t)a;
  if (p < g_blk->lds || p + 8 > g_blk->lds + LDS_BYTES) die("LDS address outside the workgroup's LDS");
  return p;
}

// ds_read_b64_tr_b16: per 16-lane group, lane t supplies the address of 4 contiguous 16-bit elements M[t][0..3]; lane t receives
// M[4 k + t / 4][t % 4], k = 0..3 (cdna_hip_programming.md: "column (l & 15) of a 4 x 16 row-major matrix per 16-lane group")
inline e_s16x4 ds_read_tr16_b64(unsigned addr) {
  if (addr & 7) die("ds_read_b64_tr_b16 address not 8-byte aligned");
  e_s16x4 out;
  wave_sync(&addr, &out, [](Wave& w) {
    t_cnt.tr_reads++;
    for (int l = 0; l < 64; l++) {
      if (!(w.amask >> l & 1)) continue;
      const int g = l & ~15, t = l & 15;
      e_s16x4 r;
      for (int k = 0; k < 4; k++) {
        const int src = g + 4 * k + t / 4;
        if (!(w.amask >> src & 1)) die("ds_read_b64_tr_b16 with a partly inactive 16-lane group");
        const char* p = lds_ptr(*(const unsigned*)w.in[src]);
        short v; memcpy(&v, p + 2 * (t % 4), 2);
        r[k] = v;
      }
      *(e_s16x4*)w.out[l] = r;
    }
  });
  return out;
}
inline e_u32x2 ds_read_tr16_b64_u32x2(unsigned addr) { return __builtin_bit_cast(e_u32x2, ds_read_tr16_b64(addr)); }

// ---- buffer resources, LDS-DMA ---------------------------------------------------------------------------------------------------------------
struct BufRsrc { const char* base; uint32_t num; };
inline BufRsrc make_rsrc(const void* p, int /*stride*/, int num, int /*flags*/) { return BufRsrc{(const char*)p, (uint32_t)num}; }
inline void buf_fetch(const BufRsrc& r, uint32_t off, int bytes, uint8_t* dst) {   // raw buffer: a dword outside [0, num) reads as 0
  for (int i = 0; i < bytes; i += 4) {
    const uint32_t o = off + (uint32_t)i;
    if (o < r.num && o + 4 <= r.num) memcpy(dst + i, r.base + o, 4); else memset(dst + i, 0, 4);
  }
}
struct DmaLaneIn { char* dst; int bytes; uint8_t data[16]; };
inline void dma_collect(Wave& w) {
  DmaOp op; op.bytes = 0; op.mask = w.amask; op.dst = nullptr; op.nop = false;
  for (int l = 0; l < 64; l++) {
    if (!(w.amask >> l & 1)) continue;
    const DmaLaneIn* in = (const DmaLaneIn*)w.in[l];
    if (!op.dst) { op.dst = in->dst; op.bytes = in->bytes; }
    if (op.dst != in->dst) die("LDS-DMA: the LDS base (M0) differs between lanes of a wave");
    memcpy(op.data[l], in->data, in->bytes);
  }
  t_cnt.dma_ops++;
  if (g_cfg.dma_late) w.dma.push_back(op); else apply_dma(op);
}
// __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, lds base (wave-uniform), size, voffset, soffset, inst offset, aux):
// lane l moves `size` bytes from rsrc[voffset + soffset + ioffset] to lds base + ioffset(?) + l * size. (Every call in this tree passes 0, 0.)
inline void buffer_load_lds(const BufRsrc& r, void* ldsbase, int size, int voff, int soff, int ioff, int /*aux*/) {
  if (size != 16 && size != 4) die("LDS-DMA size");
  DmaLaneIn in; in.dst = (char*)ldsbase + ioff; in.bytes = size;
  buf_fetch(r, (uint32_t)voff + (uint32_t)soff + (uint32_t)ioff, size, in.data);
  wave_sync(&in, nullptr, dma_collect);
}
inline void global_load_lds(const void* g, void* ldsbase, int size, int ioff, int /*aux*/) {
  DmaLaneIn in; in.dst = (char*)ldsbase + ioff; in.bytes = size;
  memcpy(in.data, (const char*)g + ioff, size);
  wave_sync(&in, nullptr, dma_collect);
}
inline e_u32x4 buffer_load_b128(const BufRsrc& r, int voff, int soff, int /*aux*/) {
  e_u32x4 v; buf_fetch(r, (uint32_t)voff + (uint32_t)soff, 16, (uint8_t*)&v);
  return v;
}
inline void buffer_store_b128(e_u32x4 v, const BufRsrc& r, int voff, int soff, int /*aux*/) {
  const uint32_t o = (uint32_t)voff + (uint32_t)soff;
  for (int i = 0; i < 4; i++)
    if (o + 4 * i < r.num && o + 4 * i + 4 <= r.num) memcpy((char*)r.base + o + 4 * i, (const char*)&v + 4 * i, 4);
}
// s_waitcnt vmcnt(n): at most n of this wave's vector-memory operations outstanding. Only LDS-DMA is queued (plain loads and stores complete
// at once here and are not counted: the model keeps MORE transfers pending than the hardware would, never fewer).
inline void waitcnt_vm(int n) {
  if (!g_cfg.dma_late) return;
  wave_sync(&n, nullptr, [](Wave& w) {
    int n0 = -1;
    for (int l = 0; l < 64; l++) if (w.amask >> l & 1) { n0 = *(const int*)w.in[l]; break; }
    drain_dma(w, (size_t)n0);
  });
}
// the lanes of a wave are independent fibres between collectives; the hardware executes them in lockstep. Where a kernel passes data between the lanes of ONE
// wave through LDS without a workgroup barrier (conv_sk.h's wave-private epilogue tile), its explicit `s_waitcnt lgkmcnt` is the re-alignment point.
inline void complete_reads(Lane* me, size_t keep) {
  while (me->pending.size() > keep) { memcpy(me->pending.front().dst, me->pending.front().val, 8); me->pending.pop_front(); }
}
// s_waitcnt lgkmcnt(n): at most n of this lane's issued LDS reads stay outstanding; the fibres of the wave re-align
inline void wave_lockstep(int n = 0) { complete_reads(g_cur, (size_t)n); int z = 0; wave_sync(&z, nullptr, [](Wave&) {}); }
// an inline-assembly transpose read: the value is fetched now (LDS contents at issue time), the destination gets it at the wait
template <class T> inline void tr_read_deferred(T& dst, unsigned addr) {
  static_assert(sizeof(T) == 8, "ds_read_b64_tr_b16 destination");
  const e_s16x4 v = ds_read_tr16_b64(addr);
  Lane::PendingRead pr; pr.dst = &dst; memcpy(pr.val, &v, 8);
  g_cur->pending.push_back(pr);
  memset(&dst, 0xFF, 8);           // poison (two bf16 NaN pairs per dword): using the register before the wait shows
}
inline void s_barrier() { block_barrier(); }
inline void syncthreads() { complete_reads(g_cur, 0); waitcnt_vm(0); block_barrier(); }      // __syncthreads() = s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier

// ---- cross-lane --------------------------------------------------------------------------------------------------------------------------------
inline int readfirstlane(int v) {
  int out;
  wave_sync(&v, &out, [](Wave& w) {
    int first = 0;
    for (int l = 0; l < 64; l++) if (w.amask >> l & 1) { first = *(const int*)w.in[l]; break; }
    for (int l = 0; l < 64; l++) if (w.amask >> l & 1) *(int*)w.out[l] = first;
  });
  return out;
}
struct ShflIn { uint64_t v; int src; };
template <class T> inline T shfl_idx(T v, int src) {
  static_assert(sizeof(T) <= 8, "shuffle payload");
  ShflIn in; in.v = 0; memcpy(&in.v, &v, sizeof(T)); in.src = src & 63;
  uint64_t out = 0;
  wave_sync(&in, &out, [](Wave& w) {
    uint64_t vals[64];
    for (int l = 0; l < 64; l++) vals[l] = (w.amask >> l & 1) ? ((const ShflIn*)w.in[l])->v : 0;
    for (int l = 0; l < 64; l++) if (w.amask >> l & 1) {
      const int s = ((const ShflIn*)w.in[l])->src;
      *(uint64_t*)w.out[l] = (w.amask >> s & 1) ? vals[s] : vals[l];      // an inactive source returns the lane's own value
    }
  });
  T r; memcpy(&r, &out, sizeof(T));
  return r;
}
inline int update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool /*bound_ctrl*/) {
  if (ctrl >= 0x100 || row_mask != 0xF || bank_mask != 0xF) die("update_dpp: only quad_perm with full masks is modelled");
  const int l = g_cur->lane;
  (void)old;
  return shfl_idx(src, (l & ~3) + ((ctrl >> (2 * (l & 3))) & 3));
}

// ---- matrix cores ------------------------------------------------------------------------------------------------------------------------------
inline float bf16_bits_to_f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
struct MfmaIn { e_bf16x8 a, b; e_f32x16 c; };
// v_mfma_f32_32x32x16_bf16: D[i][j] = C[i][j] + sum_k A[i][k] B[k][j]; lane l holds A[l % 32][8 (l / 32) .. + 7], B[8 (l / 32) .. + 7][l % 32],
// and in register r of C / D: row 8 (r / 4) + 4 (l / 32) + r % 4, column l % 32.
void mfma_bf16_fire(Wave& w);                 // defined once, in the translation unit that sets HIPEMU_IMPL (stubs.cpp: built -O3 for this loop)
inline e_f32x16 mfma_32x32x16_bf16(e_bf16x8 a, e_bf16x8 b, e_f32x16 c) {
  MfmaIn in{a, b, c};
  e_f32x16 out;
  wave_sync(&in, &out, mfma_bf16_fire);
  return out;
}
#ifdef HIPEMU_IMPL
void mfma_bf16_fire(Wave& w) {
  if (w.amask != ~0ull) die("MFMA with inactive lanes");
  t_cnt.mfma++;
  alignas(64) float A[32][16], B[16][32], D[32][32];
  for (int l = 0; l < 64; l++) {
    const MfmaIn* in = (const MfmaIn*)w.in[l];
    uint16_t ab[8], bb[8];
    memcpy(ab, &in->a, 16); memcpy(bb, &in->b, 16);
    for (int e = 0; e < 8; e++) { A[l % 32][8 * (l / 32) + e] = bf16_bits_to_f(ab[e]); B[8 * (l / 32) + e][l % 32] = bf16_bits_to_f(bb[e]); }
    for (int r = 0; r < 16; r++) D[8 * (r / 4) + 4 * (l / 32) + r % 4][l % 32] = in->c[r];
  }
  // D[i][j] += A[i][k] B[k][j], k ascending for every (i, j): one fp32 product and one fp32 sum per step (no contraction: -ffp-contract=off)
  for (int i = 0; i < 32; i++)
    for (int k = 0; k < 16; k++) {
      const float a = A[i][k];
      for (int j = 0; j < 32; j++) D[i][j] += a * B[k][j];
    }
  for (int l = 0; l < 64; l++) {
    e_f32x16 d;
    for (int r = 0; r < 16; r++) d[r] = D[8 * (r / 4) + 4 * (l / 32) + r % 4][l % 32];
    *(e_f32x16*)w.out[l] = d;
  }
}
#endif
struct Mfma2In { float a, b; e_f32x16 c; };
// v_mfma_f32_32x32x2_f32: lane l holds A[l % 32][l / 32] and B[l / 32][l % 32]
inline e_f32x16 mfma_32x32x2_f32(float a, float b, e_f32x16 c) {
  Mfma2In in{a, b, c};
  e_f32x16 out;
  wave_sync(&in, &out, [](Wave& w) {
    if (w.amask != ~0ull) die("MFMA with inactive lanes");
    t_cnt.mfma++;
    float A[32][2], B[2][32];
    for (int l = 0; l < 64; l++) { const Mfma2In* in = (const Mfma2In*)w.in[l]; A[l % 32][l / 32] = in->a; B[l / 32][l % 32] = in->b; }
    for (int l = 0; l < 64; l++) {
      const Mfma2In* in = (const Mfma2In*)w.in[l];
      e_f32x16 d;
      for (int r = 0; r < 16; r++) {
        const int i = 8 * (r / 4) + 4 * (l / 32) + r % 4;
        d[r] = in->c[r] + A[i][0] * B[0][l % 32] + A[i][1] * B[1][l % 32];
      }
      *(e_f32x16*)w.out[l] = d;
    }
  });
  return out;
}
inline float fdot2_bf16(e_bf16x2 a, e_bf16x2 b, float c, bool /*clamp*/) {
  uint16_t ab[2], bb[2];
  memcpy(ab, &a, 4); memcpy(bb, &b, 4);
  return c + bf16_bits_to_f(ab[0]) * bf16_bits_to_f(bb[0]) + bf16_bits_to_f(ab[1]) * bf16_bits_to_f(bb[1]);
}

}  // namespace hipemu

// ---- the names the kernels use ---------------------------------------------------------------------------------------------------------------------
#define threadIdx (hipemu::g_cur->tid)
#define blockIdx (hipemu::g_blk->bid)
#define blockDim (hipemu::g_bdim)
#define gridDim (hipemu::g_grid)
#define hipLaunchKernelGGL(kern, grid, block, lds, st, ...) \
  hipemu::launch(dim3(grid), dim3(block), (size_t)(lds), [&]() { kern(__VA_ARGS__); })

#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) hipemu::mfma_32x32x16_bf16(a, b, c)
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) hipemu::mfma_32x32x2_f32(a, b, c)
#define __builtin_amdgcn_make_buffer_rsrc(p, s, n, f) hipemu::make_rsrc(p, s, n, f)
#define __builtin_amdgcn_raw_ptr_buffer_load_lds hipemu::buffer_load_lds
#define __builtin_amdgcn_global_load_lds hipemu::global_load_lds
#define __builtin_amdgcn_raw_buffer_load_b128 hipemu::buffer_load_b128
#define __builtin_amdgcn_raw_buffer_store_b128 hipemu::buffer_store_b128
#define __builtin_amdgcn_readfirstlane hipemu::readfirstlane
#define __builtin_amdgcn_s_barrier hipemu::s_barrier
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)        // a scheduling hint: no effect on results
#define __builtin_amdgcn_ds_read_tr16_b64_v4i16(p) hipemu::ds_read_tr16_b64(hipemu::lds_addr(p))
#define __builtin_amdgcn_exp2f exp2f
#define __builtin_amdgcn_update_dpp hipemu::update_dpp
#define __builtin_amdgcn_fdot2_f32_bf16 hipemu::fdot2_bf16
#define __syncthreads hipemu::syncthreads

inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
template <class T> inline T __shfl_xor(T v, int m, int = 64) { return hipemu::shfl_idx(v, hipemu::g_cur->lane ^ m); }
// v_permlane32_swap_b32 vdst, vsrc: lanes 32-63 of vdst <-> lanes 0-31 of vsrc; returns {new vdst, new vsrc}
struct hipemu_u32x2 { uint32_t v[2]; uint32_t operator[](int i) const { return v[i]; } };
inline hipemu_u32x2 hipemu_permlane32_swap(uint32_t vdst, uint32_t vsrc, bool, bool) {
  const int lane = hipemu::g_cur->lane;
  const uint32_t give = lane < 32 ? vsrc : vdst;          // what this lane hands to its partner (lane ^ 32)
  const uint32_t got = hipemu::shfl_idx(give, lane ^ 32);
  hipemu_u32x2 r;
  r.v[0] = lane < 32 ? vdst : got;
  r.v[1] = lane < 32 ? got : vsrc;
  return r;
}
#define __builtin_amdgcn_permlane32_swap hipemu_permlane32_swap
template <class T> inline T __shfl(T v, int s, int = 64) { return hipemu::shfl_idx(v, s); }
template <class T> inline T __shfl_down(T v, int d, int = 64) { const int s = hipemu::g_cur->lane + d; return hipemu::shfl_idx(v, s < 64 ? s : hipemu::g_cur->lane); }
template <class T> inline T hipemu_atomic_rmw(T* p, T v, T (*op)(T, T)) {       // compare-and-swap loop on the value's bit pattern (float / double / integers)
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "atomic width");
  typedef typename std::conditional<sizeof(T) == 4, uint32_t, uint64_t>::type U;
  U* q = (U*)p;
  U old = __atomic_load_n(q, __ATOMIC_RELAXED);
  for (;;) {
    T o; memcpy(&o, &old, sizeof(T));
    const T n = op(o, v);
    U nb; memcpy(&nb, &n, sizeof(T));
    if (__atomic_compare_exchange_n(q, &old, nb, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return o;
  }
}
template <class T> inline T atomicAdd(T* p, T v) { return hipemu_atomic_rmw<T>(p, v, [](T a, T b) { return (T)(a + b); }); }
inline float __expf(float x) { return expf(x); }
inline float __fdividef(float a, float b) { return a / b; }
inline float __frcp_rn(float a) { return 1.f / a; }
inline float rsqrtf(float a) { return 1.f / sqrtf(a); }
template <class T> inline T unsafeAtomicAdd(T* p, T v) { return atomicAdd(p, v); }

// ---- the rest of the (small) runtime / intrinsic surface the translation units of csrc use -------------------------------------------------------
typedef struct hipemu_event { int dummy; }* hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t* e) { static hipemu_event ev; *e = &ev; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t = nullptr) { memset(p, v, n); return hipSuccess; }
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline float __fadd_rn(float a, float b) { return a + b; }
inline float __fsub_rn(float a, float b) { return a - b; }
inline float __fmul_rn(float a, float b) { return a * b; }
inline float __fdiv_rn(float a, float b) { return a / b; }
inline float __logf(float a) { return logf(a); }
inline long long __double_as_longlong(double d) { long long v; memcpy(&v, &d, 8); return v; }
inline double __longlong_as_double(long long v) { double d; memcpy(&d, &v, 8); return d; }
inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
inline unsigned long long __ballot(int pred) {
  unsigned long long out = 0;
  hipemu::wave_sync(&pred, &out, [](hipemu::Wave& w) {
    unsigned long long m = 0;
    for (int l = 0; l < 64; l++) if ((w.amask >> l & 1) && *(const int*)w.in[l]) m |= 1ull << l;
    for (int l = 0; l < 64; l++) if (w.amask >> l & 1) *(unsigned long long*)w.out[l] = m;
  });
  return out;
}
template <class T> inline T atomicExch(T* p, T v) { return hipemu_atomic_rmw<T>(p, v, [](T, T b) { return b; }); }
template <class T> inline T atomicMax(T* p, T v) { return hipemu_atomic_rmw<T>(p, v, [](T a, T b) { return b > a ? b : a; }); }
template <class T> inline T atomicMin(T* p, T v) { return hipemu_atomic_rmw<T>(p, v, [](T a, T b) { return b < a ? b : a; }); }
