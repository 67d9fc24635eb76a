// diag.h -- every diagnostic macro of the compress kernels, in one place.  The product build defines none of the
// SZ_MARK / SZ_TRACE / SZ_EXP_COUNT / SZ_PROF / SZ_PE switches and every macro below expands to nothing; the tools
// under tools/ (teamtrace.hip, chainprof*.hip, c3prof.hip, rolecount.hip) compile snappy_compress.hip with one of them.
// Included by snappy_compress.hip inside namespace szk, after wave_utils.h.
#pragma once

// Diagnostic build only (tools/chainprof.hip compiles this file with -DSZ_PROF): cycle stamps per
// phase, summed per block into g_prof.  Expands to nothing in the product build.
// Analysis builds only (-DSZ_MARK): comments in the ISA at the phases of the fast dense path.
#ifdef SZ_MARK
#define MARK(name) asm volatile("; MARK " name)
#else
#define MARK(name) do { } while (0)
#endif

// Diagnostic build only (tools/teamtrace.hip, -DSZ_TRACE): time line of the team kernel's turns.  The waves of
// the first blocks store (s_memtime << 28 | turn << 4 | phase) at the phase boundaries of every turn.
#ifdef SZ_TRACE
__device__ unsigned long long *g_trace;  // [traced blocks][waves][kTraceCap]
static constexpr uint32_t kTraceCap = 16384, kTraceBlocks = 8;
#define TRACE(id) do { if (team.trace && trn_ < kTraceCap) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); team.trace[trn_++] = (t_ << 28) | ((unsigned long long)(myturn & 0xffffffu) << 4) | (id); } } while (0)
#define TRACE_DECL uint32_t trn_ = 0
#define TRACE_ATTACH(team, b, wave, nw) (team).trace = (b) < kTraceBlocks && g_trace ? g_trace + ((size_t)(b) * (nw) + (wave)) * kTraceCap : nullptr
#else
#define TRACE(id) do { } while (0)
#define TRACE_DECL do { } while (0)
#define TRACE_ATTACH(team, b, wave, nw) do { } while (0)
#endif

// Timing ablations with WRONG output (-DSZ_ABLATION, selected at run time by bits of `flags` from env SZ_DEBUG_FLAGS)
#ifdef SZ_ABLATION
#define ABLATION_BREAK_IF(c) if (c) break
#define ABLATION_KEEPS(c) (c)
// What is one instruction worth?  64 independent vector instructions (4 groups of 16: no chain of dependences, so what
// they cost is issue slots, not latency) where bit `bit` of the launch's flags asks for them: 0x1000 in the lone waves'
// tile loop, 0x2000 off the token's path of a team's turn (behind the element writer), 0x4000 on it (behind the
// stale check); output unchanged (tools/padbench.py, EXPERIMENTS.md R5.4).
#define ABLATION_PAD(flags, bit) do { if ((flags) & (bit)) { uint32_t p0_ = lane, p1_ = lane, p2_ = lane, p3_ = lane; _Pragma("unroll") for (int q_ = 0; q_ < 16; q_++) { \
    asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4" : "+v"(p0_), "+v"(p1_), "+v"(p2_), "+v"(p3_) : "v"(lane)); } } } while (0)
// ... and scalar ones (the scalar port: one per ~4 clocks per SIMD whatever the number of waves): 0x10000 / 0x20000 / 0x40000
#define ABLATION_PAD_S(flags, bit) do { if ((flags) & (bit)) { uint32_t p0_ = (flags), p1_ = (flags), p2_ = (flags), p3_ = (flags); _Pragma("unroll") for (int q_ = 0; q_ < 16; q_++) { \
    asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1" : "+s"(p0_), "+s"(p1_), "+s"(p2_), "+s"(p3_) :: "scc"); } } } while (0)
#else
#define ABLATION_BREAK_IF(c) do { } while (0)
#define ABLATION_KEEPS(c) true
#define ABLATION_PAD(flags, bit) do { } while (0)
#define ABLATION_PAD_S(flags, bit) do { } while (0)
#endif

#ifdef SZ_EXP_COUNT
__device__ unsigned long long g_finish[1u << 18];  // (tools/rolecount.hip: s_memrealtime at which each block was finished; 100 MHz, one clock for the whole GPU)
#define EXP_LONE_MARK(op) ((op) | 0x80000000u)  // a block done by a lone wave
#define EXP_SINGLE_MARK(op) ((op) | 0x40000000u)  // a block done by the single kernel (tests/test_gpu_routing.py: who wrote which block)
#define EXP_ADOPT_MARK(op) ((op) | 0x20000000u)   // a block begun by a lone wave and finished by its workgroup's team (tests/test_gpu_adoption.py)
#define EXP_FINISHED(b) do { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_finish[(b) & 0x3ffffu] = t_; } while (0)  // (every lane, same address, same value)
__device__ unsigned long long g_start[1u << 18];   // ... and at which it was taken out of the queue (round 6: the shape of a launch's end)
#define EXP_STARTED(b) do { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_start[(b) & 0x3ffffu] = t_; } while (0)
#else
#define EXP_STARTED(b) do { } while (0)
#define EXP_LONE_MARK(op) (op)
#define EXP_SINGLE_MARK(op) (op)
#define EXP_ADOPT_MARK(op) (op)
#define EXP_FINISHED(b) do { } while (0)
#endif
#ifdef SZ_PROF
__device__ unsigned long long g_prof[160];  // [128,160): phases of match_length (16 sums, 16 counts), all waves
#define ML_START() unsigned long long mlt_; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(mlt_) :: "memory")
#define ML_LAP(i) do { unsigned long long t2_; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2_) :: "memory"); if (lane == 0) { atomicAdd(&g_prof[128 + (i)], t2_ - mlt_); atomicAdd(&g_prof[144 + (i)], 1ull); } mlt_ = t2_; } while (0)  // [0,64): waves with the table in LDS (32 sums, 32 counts), [64,128): in global memory
#define PROF_DECL unsigned long long pt_ = 0, pacc_[32] = {}; unsigned pcnt_[32] = {};
#define PROF_START() do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pt_) :: "memory"); } while (0)
#define PROF_LAP(i) do { unsigned long long t2_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2_) :: "memory"); pacc_[i] += t2_ - pt_; pcnt_[i]++; pt_ = t2_; } while (0)
#define PROF_LAPV(i) do { unsigned long long t2_; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2_) :: "memory"); pacc_[i] += t2_ - pt_; pcnt_[i]++; pt_ = t2_; } while (0)
#define PROF_ADD(i, v) do { pacc_[i] += (unsigned long long)(v); pcnt_[i]++; } while (0)
#define PROF_FLUSH() do { if ((threadIdx.x & 63) == 0) { const int b_ = Tab::kInLds ? 0 : 64; for (int i_ = 0; i_ < 32; i_++) { atomicAdd(&g_prof[b_ + i_], pacc_[i_]); atomicAdd(&g_prof[b_ + 32 + i_], (unsigned long long)pcnt_[i_]); } } } while (0)
#else
#define PROF_DECL
#define PROF_START() do { } while (0)
#define PROF_LAP(i) do { } while (0)
#define PROF_LAPV(i) do { } while (0)
#define PROF_ADD(i, v) do { } while (0)
#define PROF_FLUSH() do { } while (0)
#define ML_START() do { } while (0)
#define ML_LAP(i) do { } while (0)
#endif
#if defined(SZ_PE) || defined(SZ_PROF)
__device__ unsigned long long g_pe[65536 * 4];
#endif
#if defined(SZ_PE) || defined(SZ_PROF)
// block phases (tools/c3prof.hip): 28 entry -> stream requested, 29 -> main loop, 30 main loop, 31 -> return; s_memtime (shader clock)
#define PE_START() unsigned long long pe_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pe_) :: "memory")
#define PE_LAP(i) do { unsigned long long t2_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2_) :: "memory"); if (lane == 0) g_pe[(blockIdx.x & 0xffffu) * 4u + ((i) - 28)] = t2_ - pe_; pe_ = t2_; } while (0)  // (one slot per block and phase: plain stores, nothing shared)
#else
#define PE_START() do { } while (0)
#define PE_LAP(i) do { } while (0)
#endif

