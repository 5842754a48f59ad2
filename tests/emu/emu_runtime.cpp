// TEST INFRASTRUCTURE -- the lane-by-lane executor behind tests/emu/hip/hip_runtime.h.
//
// One workgroup at a time; every thread of it is a fibre with its own stack.  A fibre runs until it exits or reaches a
// rendezvous: `__syncthreads` (all live threads of the workgroup) or a wave collective (all live lanes of its wave:
// shuffle, ballot, readlane, DPP, swizzle, MFMA, wave_barrier).  The lane that arrives last computes the results of all
// lanes (the other fibres are suspended inside the collective, so pointers into their stacks are valid) and wakes them.
//
// Checks that only an emulator can make:
//  * a collective whose lanes did NOT all arrive (the rest of the wave sits in another collective, in a barrier, or has
//    left the kernel) is completed with the lanes that did -- as the hardware does under a partial EXEC mask -- and
//    COUNTED; a lane that then reads the value of an absent lane gets a poison pattern and the event is counted as
//    `emu_reads_of_inactive_lanes` (on gfx950 such a read returns whatever the crossbar held: round 2's Leiden bug);
//  * everything the kernels index is ordinary host memory: build with EMU_ASAN=1 and AddressSanitizer sees every access.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <sys/mman.h>
#include <unistd.h>

#include <vector>

#if defined(__SANITIZE_ADDRESS__) || (defined(__has_feature) && __has_feature(address_sanitizer))
#define EMU_HAVE_ASAN 1
extern "C" void __sanitizer_start_switch_fiber(void** fake_stack_save, const void* bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void* fake_stack_save, const void** bottom_old, size_t* size_old);
#else
#define EMU_HAVE_ASAN 0
#endif

extern "C" void emu_switch(void** save_sp, void* new_sp);
extern "C" void __asan_poison_memory_region(void const volatile* addr, size_t size);
extern "C" void __asan_unpoison_memory_region(void const volatile* addr, size_t size);

namespace emu {

Lane* g_cur = nullptr;
int g_dma_late = 0;

namespace {
enum State { RUNNABLE = 0, WAIT_WAVE = 1, WAIT_BLOCK = 2, DONE = 3 };
constexpr int MAX_THREADS = 1024;
constexpr size_t STACK_BYTES = 512 << 10;
constexpr unsigned long long POISON = 0xBAD0BAD0BAD0BAD0ull;

long long g_seq = 0;
struct Stats {
  long long launches = 0, partial_collectives = 0, mixed_collectives = 0, reads_of_inactive = 0;
} g_stats;

Lane g_lanes[MAX_THREADS];
char* g_stacks = nullptr;
void* g_sched_sp = nullptr;
int g_nthreads = 0, g_nwaves = 0, g_live_block = 0, g_barrier_arrived = 0;
int g_wave_live[MAX_THREADS / 64], g_wave_arrived[MAX_THREADS / 64];
void (*g_tramp)(void*) = nullptr;
void* g_closure = nullptr;
std::vector<unsigned char> g_dyn_lds;
#if EMU_HAVE_ASAN
void* g_fake_sched = nullptr;
const void* g_sched_bottom = nullptr;
size_t g_sched_size = 0;
#endif

__asm__(
    ".text\n.globl emu_switch\n.type emu_switch,@function\nemu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n"
    ".size emu_switch,.-emu_switch\n");

void to_scheduler() {
  Lane* me = g_cur;
#if EMU_HAVE_ASAN
  void* fake = nullptr;
  __sanitizer_start_switch_fiber(me->state == DONE ? nullptr : &fake, g_sched_bottom, g_sched_size);
#endif
  emu_switch(&me->sp, g_sched_sp);
#if EMU_HAVE_ASAN
  __sanitizer_finish_switch_fiber(fake, &g_sched_bottom, &g_sched_size);
#endif
}

void complete_wave(int w, bool partial, int leader);
bool try_complete_full(int w, int leader);
void release_block_barrier();

void fibre_main() {
#if EMU_HAVE_ASAN
  __sanitizer_finish_switch_fiber(nullptr, &g_sched_bottom, &g_sched_size);
#endif
  g_tramp(g_closure);
  dma_land(0);  // (a wave does not end with memory operations outstanding)
  Lane* me = g_cur;
  me->state = DONE;
  --g_live_block;
  const int w = me->wave;
  --g_wave_live[w];
  // the lanes that stay may have been waiting for this one
  if (g_wave_live[w] > 0 && g_wave_arrived[w] == g_wave_live[w]) try_complete_full(w, -1);
  if (g_live_block > 0 && g_barrier_arrived == g_live_block) release_block_barrier();
  to_scheduler();
  abort();  // a finished fibre is never resumed
}

void init_fibre(Lane* l, int idx) {
  char* top = g_stacks + (size_t)(idx + 1) * STACK_BYTES;
#if EMU_HAVE_ASAN
  // the frames a finished fibre never returned from (fibre_main, to_scheduler) left their redzones in the shadow
  __asan_unpoison_memory_region(top - (16 << 10), 16 << 10);
#endif
  void** sp = reinterpret_cast<void**>(reinterpret_cast<uintptr_t>(top) & ~(uintptr_t)15);
  // layout popped by emu_switch: r15 r14 r13 r12 rbx rbp, then `ret` into fibre_main with rsp = 8 (mod 16)
  *--sp = nullptr;  // keeps the alignment the ABI expects at function entry
  *--sp = reinterpret_cast<void*>(&fibre_main);
  for (int i = 0; i < 6; ++i) *--sp = nullptr;
  l->sp = sp;
}

void resume(Lane* l) {
  g_cur = l;
#if EMU_HAVE_ASAN
  char* bottom = g_stacks + (size_t)(l - g_lanes) * STACK_BYTES;
  __sanitizer_start_switch_fiber(&g_fake_sched, bottom, STACK_BYTES);
#endif
  emu_switch(&g_sched_sp, l->sp);
#if EMU_HAVE_ASAN
  __sanitizer_finish_switch_fiber(g_fake_sched, nullptr, nullptr);
#endif
  g_cur = nullptr;
}

// all lanes of wave w that wait in a wave collective and share (kind, fn, site) with lane `leader` (-1: the first
// waiting lane)
void complete_wave(int w, bool partial, int leader = -1) {
  Lane* base = g_lanes + w * 64;
  const int n = std::min(64, g_nthreads - w * 64);
  int first = leader;
  if (first < 0)
    for (int l = 0; l < n; ++l)
      if (base[l].state == WAIT_WAVE) { first = l; break; }
  if (first < 0) return;
  Arrived arr[64];
  unsigned long long active = 0;
  bool mixed = false;
  for (int l = 0; l < 64; ++l) arr[l] = Arrived{nullptr, nullptr};
  for (int l = 0; l < n; ++l) {
    if (base[l].state != WAIT_WAVE) continue;
    if (base[l].kind != base[first].kind || base[l].fn != base[first].fn || base[l].site != base[first].site) {
      mixed = mixed || (base[l].kind < K_SOFT_RMW && base[first].kind < K_SOFT_RMW);
      continue;
    }
    arr[l] = base[l].arr;
    active |= 1ull << l;
  }
  if (base[first].kind < K_SOFT_RMW) {
    if (partial) ++g_stats.partial_collectives;
    if (mixed) ++g_stats.mixed_collectives;
    static int trace = getenv("EMU_TRACE_PARTIAL") ? atoi(getenv("EMU_TRACE_PARTIAL")) : 0;
    if ((partial || mixed) && trace > 0) {
      --trace;
      Dl_info di{};
      dladdr(base[first].site, &di);
      fprintf(stderr, "emu: %s%s collective kind %d at %s+0x%lx, wave %d of workgroup (%u,%u,%u), lanes %016llx; the others:",
              partial ? "partial " : "", mixed ? "mixed " : "", base[first].kind, di.dli_fname ? di.dli_fname : "?",
              (unsigned long)((const char*)base[first].site - (const char*)di.dli_fbase), w, base[first].bidx.x,
              base[first].bidx.y, base[first].bidx.z, active);
      for (int l = 0; l < n; ++l)
        if (!(active >> l & 1))
          fprintf(stderr, " %d:%s", l,
                  base[l].state == DONE ? "done" : base[l].state == WAIT_BLOCK ? "barrier" : base[l].state == WAIT_WAVE ? "other-site" : "runnable");
      for (int l = 0; l < n; ++l)
        if (!(active >> l & 1) && base[l].state == WAIT_WAVE) {
          fprintf(stderr, " [lane %d waits in kind %d at +0x%lx]", l, base[l].kind,
                  (unsigned long)((const char*)base[l].site - (const char*)di.dli_fbase));
          break;
        }
      fprintf(stderr, "\n");
    }
  }
  base[first].fn(active, arr, base[first].uniform);
  for (int l = 0; l < n; ++l)
    if (active >> l & 1) {
      base[l].state = RUNNABLE;
      --g_wave_arrived[w];
    }
}

// complete the collective lane `leader` of wave w waits in if EVERY live lane of the wave waits in the same one
bool try_complete_full(int w, int leader) {
  Lane* base = g_lanes + w * 64;
  const int n = std::min(64, g_nthreads - w * 64);
  if (leader < 0)
    for (int l = 0; l < n && leader < 0; ++l)
      if (base[l].state == WAIT_WAVE) leader = l;
  if (leader < 0) return false;
  int same = 0;
  for (int l = 0; l < n; ++l)
    same += base[l].state == WAIT_WAVE && base[l].kind == base[leader].kind && base[l].fn == base[leader].fn &&
            base[l].site == base[leader].site;
  if (same != g_wave_live[w]) return false;
  complete_wave(w, false, leader);
  return true;
}

void release_block_barrier() {
  for (int t = 0; t < g_nthreads; ++t)
    if (g_lanes[t].state == WAIT_BLOCK) g_lanes[t].state = RUNNABLE;
  g_barrier_arrived = 0;
}
}  // namespace

void wave_collective(int kind, const void* opnd, void* res, ComputeAll fn, const void* uniform) {
  Lane* me = g_cur;
  me->kind = kind;
  me->fn = fn;
  me->uniform = uniform;
  me->site = __builtin_return_address(0);
  me->arr = Arrived{opnd, res};
  me->state = WAIT_WAVE;
  me->seq = ++g_seq;
  const int w = me->wave;
  if (++g_wave_arrived[w] == g_wave_live[w] && try_complete_full(w, me->lane)) return;
  // (lanes of the wave sit in different collectives, or have not arrived: the scheduler sorts it out)
  to_scheduler();
}

void block_barrier() {
  Lane* me = g_cur;
  me->state = WAIT_BLOCK;
  me->seq = ++g_seq;
  if (++g_barrier_arrived == g_live_block) {
    release_block_barrier();
    return;
  }
  to_scheduler();
}

void* dyn_lds() { return g_dyn_lds.data(); }

static void segv_backtrace(int sig, siginfo_t* si, void*) {
  void* frames[48];
  const int n = backtrace(frames, 48);
  dprintf(2, "emu: signal %d at address %p, lane %d of wave %d, workgroup (%u,%u,%u)\n", sig, si->si_addr,
          g_cur ? g_cur->lane : -1, g_cur ? g_cur->wave : -1, g_cur ? g_cur->bidx.x : 0, g_cur ? g_cur->bidx.y : 0,
          g_cur ? g_cur->bidx.z : 0);
  backtrace_symbols_fd(frames, n, 2);
  _exit(139);
}

void launch_body(dim3 grid, dim3 block, size_t shmem, void (*tramp)(void*), void* closure) {
  static bool handler = false;
  if (!handler && getenv("EMU_SEGV_BACKTRACE")) {
    handler = true;
    static char alt[1 << 16];
    stack_t ss{alt, 0, sizeof(alt)};
    sigaltstack(&ss, nullptr);
    struct sigaction sa {};
    sa.sa_sigaction = segv_backtrace;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
    sigaction(SIGSEGV, &sa, nullptr);
    sigaction(SIGBUS, &sa, nullptr);
  }
  const int nthreads = (int)(block.x * block.y * block.z);
  if (nthreads <= 0 || nthreads > MAX_THREADS) {
    fprintf(stderr, "emu: workgroup of %d threads\n", nthreads);
    abort();
  }
  if (!g_stacks) {
    g_stacks = static_cast<char*>(mmap(nullptr, (size_t)MAX_THREADS * STACK_BYTES, PROT_READ | PROT_WRITE,
                                       MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
    if (g_stacks == MAP_FAILED) abort();
  }
  ++g_stats.launches;
  g_tramp = tramp;
  g_closure = closure;
  // 16-byte aligned dynamic LDS, poisoned per workgroup (LDS is not zero on the hardware either)
  g_dyn_lds.assign(shmem + 64, 0xCD);
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        if (shmem) memset(g_dyn_lds.data(), 0xCD, shmem);
        g_nthreads = nthreads;
        g_nwaves = (nthreads + 63) / 64;
        g_live_block = nthreads;
        g_barrier_arrived = 0;
        for (int w = 0; w < g_nwaves; ++w) {
          g_wave_live[w] = std::min(64, nthreads - w * 64);
          g_wave_arrived[w] = 0;
        }
        for (int t = 0; t < nthreads; ++t) {
          Lane& l = g_lanes[t];
          l.tidx = Idx3{t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
          l.bidx = Idx3{bx, by, bz};
          l.bdim = Idx3{block.x, block.y, block.z};
          l.gdim = Idx3{grid.x, grid.y, grid.z};
          l.flat = t;
          l.wave = t >> 6;
          l.lane = t & 63;
          l.state = RUNNABLE;
          l.n_dma = 0;
          init_fibre(&l, t);
        }
        while (g_live_block > 0) {
          bool ran = false;
          for (int w = 0; w < g_nwaves; ++w) {
            bool again = true;
            while (again) {
              again = false;
              const int n = std::min(64, nthreads - w * 64);
              for (int l = 0; l < n; ++l) {
                Lane& ln = g_lanes[w * 64 + l];
                if (ln.state == RUNNABLE) {
                  resume(&ln);
                  ran = again = true;
                }
              }
            }
          }
          if (ran) continue;
          // nothing can run: some wave holds lanes in a collective the rest of it will not reach (they wait in the
          // workgroup barrier or in another collective) -- the hardware executes it with the lanes that are there
          // Lock-step points first: pending read-modify-writes, then pending atomic loads (a lane that skipped a loop
          // of inserts waits at the load behind the loop until the lanes inside it are done -- the hardware's
          // reconvergence at the loop exit); within a class the lowest call site (deterministic; the order of two
          // independent read-modify-writes does not matter).
          bool resolved = false;
          int best_t = -1;
          for (int pass_kind : {(int)K_SOFT_RMW, (int)K_SOFT_LOAD}) {
            for (int t = 0; t < nthreads; ++t)
              if (g_lanes[t].state == WAIT_WAVE && g_lanes[t].kind == pass_kind &&
                  (best_t < 0 || g_lanes[t].site < g_lanes[best_t].site))
                best_t = t;
            if (best_t >= 0) break;
          }
          if (best_t >= 0) {
            complete_wave(best_t >> 6, true, best_t & 63);
            resolved = true;
          }
          // no lock-step point is pending: a shuffle / ballot / ... that part of its wave will not reach (counted)
          for (int w = 0; w < g_nwaves && !resolved; ++w)
            if (g_wave_arrived[w] > 0) {
              complete_wave(w, true, -1);
              resolved = true;
            }
          if (!resolved) {
            fprintf(stderr, "emu: deadlock in workgroup (%u,%u,%u): %d live threads, %d in the barrier\n", bx, by, bz,
                    g_live_block, g_barrier_arrived);
            abort();
          }
        }
      }
}

// ---- results of the collectives ------------------------------------------------------------------------------------
static inline void set64(void* res, unsigned long long v) { memcpy(res, &v, 8); }

void shfl_all(unsigned long long active, const Arrived* l, const void*) {
  unsigned long long out[64];
  for (int i = 0; i < 64; ++i) {
    if (!(active >> i & 1)) continue;
    const ShflOp* op = static_cast<const ShflOp*>(l[i].opnd);
    const int s = op->src < 0 ? i : op->src & 63;
    if (active >> s & 1) out[i] = static_cast<const ShflOp*>(l[s].opnd)->val;
    else {
      out[i] = POISON;
      ++g_stats.reads_of_inactive;
    }
  }
  for (int i = 0; i < 64; ++i)
    if (active >> i & 1) set64(l[i].res, out[i]);
}
void readlane_all(unsigned long long active, const Arrived* l, const void*) {
  unsigned long long out[64];
  for (int i = 0; i < 64; ++i) {
    if (!(active >> i & 1)) continue;
    const int s = static_cast<const ShflOp*>(l[i].opnd)->src & 63;
    // (v_readlane reads the register of lane s whether or not it is enabled: a lane that has LEFT the kernel or sits
    // elsewhere still has no meaningful value)
    if (active >> s & 1) out[i] = static_cast<const ShflOp*>(l[s].opnd)->val;
    else {
      out[i] = POISON;
      ++g_stats.reads_of_inactive;
    }
  }
  for (int i = 0; i < 64; ++i)
    if (active >> i & 1) set64(l[i].res, out[i]);
}
void readfirst_all(unsigned long long active, const Arrived* l, const void*) {
  const int f = __builtin_ctzll(active);
  unsigned long long v;
  memcpy(&v, l[f].opnd, 8);
  for (int i = 0; i < 64; ++i)
    if (active >> i & 1) set64(l[i].res, v);
}
void ballot_all(unsigned long long active, const Arrived* l, const void*) {
  unsigned long long m = 0;
  for (int i = 0; i < 64; ++i)
    if ((active >> i & 1) && *static_cast<const int*>(l[i].opnd)) m |= 1ull << i;
  for (int i = 0; i < 64; ++i)
    if (active >> i & 1) set64(l[i].res, m);
}
void barrier_all(unsigned long long, const Arrived*, const void*) {}
void thunks_all(unsigned long long active, const Arrived* l, const void*) {
  for (int i = 0; i < 64; ++i)
    if (active >> i & 1) {
      const Thunk* t = static_cast<const Thunk*>(l[i].opnd);
      t->run(t->ctx);
    }
}

// D = A x B + C on the 64 lanes of a wave; register layouts of the CDNA3/4 ISA guide (MI355X_MICROARCH.md):
//  32x32 results: lane l holds column j = l % 32, register r holds row i = 8 * (r / 4) + 4 * (l / 32) + r % 4
static void require_full(unsigned long long active, const char* what) {
  if (active != ~0ull) {
    fprintf(stderr, "emu: %s executed by a partial wave (mask %016llx)\n", what, active);
    abort();
  }
}
void mfma_f32_32x32x2_all(unsigned long long active, const Arrived* l, const void*) {
  require_full(active, "v_mfma_f32_32x32x2_f32");
  float a[32][2], b[2][32];
  for (int i = 0; i < 64; ++i) {
    const MfmaF32Op* op = static_cast<const MfmaF32Op*>(l[i].opnd);
    a[i % 32][i / 32] = op->a;  // A: row i % 32, k = i / 32
    b[i / 32][i % 32] = op->b;  // B: k = i / 32, column i % 32
  }
  for (int i = 0; i < 64; ++i) {
    const MfmaF32Op* op = static_cast<const MfmaF32Op*>(l[i].opnd);
    f32x16_t d = op->c;
    const int j = i % 32;
    for (int r = 0; r < 16; ++r) {
      const int row = 8 * (r / 4) + 4 * (i / 32) + r % 4;
      float acc = d[r];
      for (int k = 0; k < 2; ++k) acc = fmaf(a[row][k], b[k][j], acc);
      d[r] = acc;
    }
    memcpy(l[i].res, &d, sizeof(d));
  }
}
static inline float bf16_to_f32(unsigned short h) {
  unsigned u = (unsigned)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
void mfma_bf16_32x32x16_all(unsigned long long active, const Arrived* l, const void*) {
  require_full(active, "v_mfma_f32_32x32x16_bf16");
  static float a[32][16], b[16][32];
  for (int i = 0; i < 64; ++i) {
    const MfmaBf16Op* op = static_cast<const MfmaBf16Op*>(l[i].opnd);
    for (int q = 0; q < 8; ++q) {
      a[i % 32][8 * (i / 32) + q] = bf16_to_f32(op->a[q]);  // A: row i % 32, k = 8 * (i / 32) + q
      b[8 * (i / 32) + q][i % 32] = bf16_to_f32(op->b[q]);  // B: column i % 32, same k
    }
  }
  for (int i = 0; i < 64; ++i) {
    const MfmaBf16Op* op = static_cast<const MfmaBf16Op*>(l[i].opnd);
    f32x16_t d = op->c;
    const int j = i % 32;
    for (int r = 0; r < 16; ++r) {
      const int row = 8 * (r / 4) + 4 * (i / 32) + r % 4;
      double acc = 0.0;  // (the products of bf16 pairs are exact in float32; the hardware's internal sum is wider than float32)
      for (int k = 0; k < 16; ++k) acc += (double)a[row][k] * (double)b[k][j];
      d[r] = (float)((double)d[r] + acc);
    }
    memcpy(l[i].res, &d, sizeof(d));
  }
}
void mfma_f64_16x16x4_all(unsigned long long active, const Arrived* l, const void*) {
  require_full(active, "v_mfma_f64_16x16x4_f64");
  double a[16][4], b[4][16];
  for (int i = 0; i < 64; ++i) {
    const MfmaF64Op* op = static_cast<const MfmaF64Op*>(l[i].opnd);
    a[i % 16][i / 16] = op->a;  // A: row i % 16, k = i / 16
    b[i / 16][i % 16] = op->b;  // B: k = i / 16, column i % 16
  }
  for (int i = 0; i < 64; ++i) {
    const MfmaF64Op* op = static_cast<const MfmaF64Op*>(l[i].opnd);
    f64x4_t d = op->c;
    const int j = i % 16;
    for (int r = 0; r < 4; ++r) {
      const int row = (i / 16) + 4 * r;  // D: lane l holds column l % 16, rows (l / 16) + 4 r (pinned on the GPU by
                                         // tests/test_gpu_dense.py::test_dgemm_tn; NOT the float32 16x16x4 layout)
      double acc = d[r];
      for (int k = 0; k < 4; ++k) acc = fma(a[row][k], b[k][j], acc);
      d[r] = acc;
    }
    memcpy(l[i].res, &d, sizeof(d));
  }
}

}  // namespace emu

extern "C" {
// gaps between the buffers carved from a workspace (tests/emu/build.py rewrites Workspace::take for the ASan build)
void emu_poison_gap(void* p, size_t n) {
#if EMU_HAVE_ASAN
  if (n) __asan_poison_memory_region(p, n);
#endif
}
void emu_unpoison(void* p, size_t n) {
#if EMU_HAVE_ASAN
  if (n) __asan_unpoison_memory_region(p, n);
#endif
}
long long emu_launches() { return emu::g_stats.launches; }
long long emu_partial_collectives() { return emu::g_stats.partial_collectives; }
long long emu_mixed_collectives() { return emu::g_stats.mixed_collectives; }
long long emu_reads_of_inactive_lanes() { return emu::g_stats.reads_of_inactive; }
// event counters a kernel may bump under `#ifdef SCAMD_EMU` (tests/emu/hip/hip_runtime.h declares the array): what the
// hardware counters cannot say, e.g. how many list insertions a query of the kNN sweep costs (tools/emu_knn_insertions.py)
void emu_set_dma_late(int late) { emu::g_dma_late = late; }
long long emu_user_counters[16] = {0};
long long emu_user_counter(int i) { return i >= 0 && i < 16 ? emu_user_counters[i] : -1; }
void emu_reset_stats() {
  emu::g_stats = emu::Stats{};
  for (long long& c : emu_user_counters) c = 0;
}
}
