// snappy_abi.cpp -- the C ABI of include/snappy_hip.h over the gfx950 kernels.
//
// Host side of the drop-in boundary.  Mirrors, for the one hot path, the call surfaces of the
// reference: the RoCC command port driven by tests/compression.c:20-39 (decoded in
// CompressionAccelerator.scala:54-73,361-389) and snappy::Compress / snappy::Uncompress as called
// by tests/swcompression.cc:27,31,45.  There is NO CPU compression path in this library: every
// compute entry point runs the HIP kernels or fails with SZ_ENODEV / SZ_EHIP.
#include "snappy_hip.h"

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <memory>
#include <mutex>
#include <utility>
#include <vector>

#include "snappy_kernels.h"

namespace {

// Multi-wave kernel workspace: one per (device, stream), created when a launch first has enough blocks
// for the multi-wave kernel and kept until the stream's resources are released (64 MiB on MI355X:
// 2 048 tables of 32 KiB).  SZ_KERNEL=single disables the dual kernel.
// The workspace holds the launch state of ONE launch sequence at a time (sample counters, block
// queue): `launch_mu` is held while a call enqueues its reset / sample / single / dual kernels, so
// that two host threads launching on the same stream (e.g. both on the NULL stream) cannot
// interleave those sequences -- on the stream the kernels of one call then run back to back.
struct DualSlot {
    szk::DualScratch sc{nullptr, nullptr, nullptr, 0, 0, 0, szk::kDualAuto, szk::kKernelTeam};
    std::mutex launch_mu;
    bool retired = false;  // under launch_mu: the buffers are gone (drop_dual_scratch); a launcher that still holds the slot takes the single kernel
    DualSlot() = default;
    DualSlot(const DualSlot &) = delete;
    DualSlot &operator=(const DualSlot &) = delete;
    // No destructor: device memory is freed on the explicit release path only (drop_dual_scratch), never by
    // whoever happens to hold the slot last and never at static destruction / dlclose, when the HIP runtime may
    // already be gone (the same policy as the per-thread workspace below).
};
// The map and its mutex are created on first use and never destroyed, so that no destructor of this library
// runs HIP calls (or touches the slots) at process exit; workspaces still cached then are left to the driver.
std::mutex &dual_mutex()
{
    static std::mutex *m = new std::mutex;
    return *m;
}
std::map<std::pair<int, hipStream_t>, std::shared_ptr<DualSlot>> &dual_map()
{
    static auto *m = new std::map<std::pair<int, hipStream_t>, std::shared_ptr<DualSlot>>;
    return *m;
}

int kernel_mode()
{
    // default (auto): when there are enough blocks to give every SIMD two waves, a sample of the data decides
    // between the single kernel (incompressible) and the team kernel (compressible).  SZ_KERNEL=single: always
    // the single kernel; team: always the team kernel, from two blocks up; dual: always the multi-wave kernel
    // (round 1-2's kernel for compressible data, kept as a second implementation for differential tests and A/B),
    // from two blocks up.  Read on every call so that a test can switch modes inside one process.
    const char *e = getenv("SZ_KERNEL");
    return !e ? 0 : strcmp(e, "single") == 0 ? 1 : strcmp(e, "dual") == 0 ? 2 : strcmp(e, "team") == 0 ? 4 : 0;
}

// The workspace slot of (current device, stream) for a launch of n_blocks, or nullptr when the
// single kernel is to run (mode, too few blocks, no memory, stream being captured before a first use).
// `view` receives the launch parameters for this call.
// (The slot is returned as a shared_ptr: a concurrent sz_release_stream_resources / Workspace::release of the
// same stream removes it from the map and frees its buffers under launch_mu; a launcher that still holds the
// slot sees `retired` once it has the lock -- lock_live_slot -- and never enqueues on freed memory.)  Launches
// that the small-block kernel takes never use the workspace.
std::shared_ptr<DualSlot> dual_slot_for(hipStream_t stream, uint32_t n_blocks, size_t in_stride, const uint32_t *d_in_len,
                                        szk::DualScratch *view)
{
    const int mode = kernel_mode();
    if (mode == 1)
        return nullptr;
    if (!d_in_len && in_stride <= 8192)  // fixed small blocks: compress_small_blocks_kernel (launch_any), no workspace
        return nullptr;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess)
        return nullptr;
    std::lock_guard<std::mutex> lock(dual_mutex());
    auto &g_dual = dual_map();
    auto key = std::make_pair(dev, stream);
    auto it = g_dual.find(key);
    if (it == g_dual.end()) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            return nullptr;
        const uint32_t n_slots = 4u * (uint32_t)cus;  // four workgroups per CU, kGroupWaves waves each
        // (sampled launches from n_slots + 1 blocks up: with at most one block per SIMD the single kernel is as fast on any
        // data -- one wave alone on its SIMD --, from the second round on the persistent kernel wins on compressible data:
        // 1 536 text blocks 3.45 -> 2.63 ms, round 5; rounds 1-4: from 2 x n_slots)
        const uint32_t min_blocks = (mode == 2 || mode == 4) ? 2u : n_slots + 1u;
        if (n_blocks < min_blocks)
            return nullptr;  // (a 130-byte sz_compress never pays for the 64 MiB workspace)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
            (void)hipGetLastError();
            return nullptr;  // no allocation while a graph is being captured: single kernel this time
        }
        std::shared_ptr<DualSlot> slot(new DualSlot);
        slot->sc.n_slots = n_slots;
        void *t = nullptr, *g = nullptr, *q = nullptr;
        const size_t gbytes = szk::kLaunchStateBytes;  // launch state (+ the sample's per-workgroup results)
        const size_t tbytes = (size_t)n_slots * szk::kDualTablesPerGroup * szk::kDualTableBytes;
        const size_t qbytes = (size_t)szk::kDualQueueEntries * sizeof(uint32_t);
        if (hipMalloc(&t, tbytes) == hipSuccess && hipMalloc(&g, gbytes) == hipSuccess &&
            hipMalloc(&q, qbytes) == hipSuccess && hipMemsetAsync(t, 0, tbytes, stream) == hipSuccess &&
            hipMemsetAsync(g, 0, gbytes, stream) == hipSuccess) {
            slot->sc.tables = static_cast<uint16_t *>(t);  // (never cleared again: a bitmap in LDS says which buckets a block has written; the memset is hygiene)
            slot->sc.select = static_cast<uint32_t *>(g);
            slot->sc.queue = static_cast<uint32_t *>(q);
            slot->sc.queue_cap = szk::kDualQueueEntries;
        } else {
            (void)hipGetLastError();
            if (t) (void)hipFree(t);
            if (g) (void)hipFree(g);
            if (q) (void)hipFree(q);
            return nullptr;  // tried again by the next large launch
        }
        it = g_dual.emplace(key, std::move(slot)).first;
    }
    std::shared_ptr<DualSlot> s = it->second;
    *view = s->sc;
    view->mode = (mode == 2 || mode == 4) ? szk::kDualAlways : szk::kDualAuto;
    // compressible launches: the team kernel (a two-wave team on the LDS table + two global-table waves per
    // workgroup), unless the multi-wave kernel is asked for by name (SZ_KERNEL=dual)
    view->kernel = mode == 2 ? szk::kKernelMulti : szk::kKernelTeam;
    view->min_blocks = (mode == 2 || mode == 4) ? 2u : s->sc.n_slots + 1u;
    if (n_blocks < view->min_blocks)
        return nullptr;
    return s;
}

// Frees the multi-wave workspace that belongs to a stream which is about to be destroyed.
void drop_dual_scratch(int dev, hipStream_t stream)
{
    std::shared_ptr<DualSlot> slot;
    {
        std::lock_guard<std::mutex> lock(dual_mutex());
        auto &g_dual = dual_map();
        auto it = g_dual.find(std::make_pair(dev, stream));
        if (it == g_dual.end())
            return;
        slot = std::move(it->second);
        g_dual.erase(it);
    }
    // Freed here, by the caller that releases the stream's resources (it has synchronised the stream), under
    // launch_mu: no launch sequence is half enqueued, and a launcher that looked the slot up a moment ago finds
    // it retired once it holds the lock and runs the single kernel instead (lock_live_slot).
    std::lock_guard<std::mutex> launch(slot->launch_mu);
    slot->retired = true;
    if (slot->sc.tables)
        (void)hipFree(slot->sc.tables);
    if (slot->sc.select)
        (void)hipFree(slot->sc.select);
    if (slot->sc.queue)
        (void)hipFree(slot->sc.queue);
    slot->sc.tables = nullptr;
    slot->sc.select = nullptr;
    slot->sc.queue = nullptr;
}

// Takes the launch lock of a looked-up slot; a slot retired in between is let go (single kernel this time).
std::unique_lock<std::mutex> lock_live_slot(std::shared_ptr<DualSlot> &slot)
{
    std::unique_lock<std::mutex> seq;
    if (slot) {
        seq = std::unique_lock<std::mutex>(slot->launch_mu);
        if (slot->retired) {
            seq.unlock();
            seq = std::unique_lock<std::mutex>();
            slot.reset();
        }
    }
    return seq;
}

thread_local int tl_hip_error = 0;
thread_local int tl_rocc_error = 0;
thread_local uint32_t tl_rocc_length = 0;  // the accelerator's latched `length` register

inline int hip_fail(hipError_t e)
{
    tl_hip_error = (int)e;
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver)
        return SZ_ENODEV;
    return SZ_EHIP;
}
#define SZ_HIP(call)                   \
    do {                               \
        hipError_t e_ = (call);        \
        if (e_ != hipSuccess)          \
            return hip_fail(e_);       \
    } while (0)

// The caller's (pageable) buffer made DMA-able in place for the length of one call.  A copy to or from pageable memory
// goes through the runtime's own staging buffers on the calling thread -- it is not asynchronous, and the copy in and
// the copy out of one call cannot overlap; a registered range is copied by the DMA engines alone (tools/hostxfer.hip:
// 57 GB/s each way, 49 both at once; registering costs less than the copy it enables).  Registration can fail -- a
// read-only mapping, a range the caller has registered already, a platform without it --: then `ok` is false and the
// copies are the runtime's pageable ones, correct and slower.  Always released before the call returns: a registration
// must not outlive the caller's right to the memory.
struct HostPin {
    void *p = nullptr;
    bool ok = false;
    HostPin(const void *ptr, size_t n, bool read_only)
    {
        if (!ptr || n == 0 || getenv("SZ_NO_HOST_REGISTER"))
            return;
        void *q = const_cast<void *>(ptr);
        hipError_t e = hipHostRegister(q, n, hipHostRegisterDefault);
        if (e != hipSuccess && read_only) {
            (void)hipGetLastError();
            e = hipHostRegister(q, n, hipHostRegisterReadOnly);
        }
        if (e == hipSuccess) {
            p = q;
            ok = true;
        } else {
            (void)hipGetLastError();
        }
    }
    ~HostPin()
    {
        if (ok)
            (void)hipHostUnregister(p);
    }
    HostPin(const HostPin &) = delete;
    HostPin &operator=(const HostPin &) = delete;
};

// Grow-only device workspace of the host-buffer API, one per host thread (one RoCC port per hart).
struct Workspace {
    hipStream_t stream = nullptr;
    void *buf[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t cap[5] = {0, 0, 0, 0, 0};
    int device = -1;
    std::vector<uint8_t> host;  // staging of the one-fragment path (slot + length in one copy)
    // the pipeline of large host buffers (compress_pipelined): copies in and out on streams of their own, one event per
    // chunk and direction, the chunks' running totals in pinned memory
    hipStream_t s_in = nullptr, s_out = nullptr;
    std::vector<hipEvent_t> ev;
    uint64_t *pin_tot = nullptr;
    static constexpr size_t kPinTot = 64;
    // a call between sz_compress_begin and sz_compress_finish (sz_compress is the two back to back)
    struct Pending {
        bool active = false, enqueued = false;
        const uint8_t *src = nullptr;
        size_t n = 0, nh = 0;
        uint8_t hdr[5] = {0, 0, 0, 0, 0};
        uint32_t stages = 0;
        std::unique_ptr<HostPin> pin_src;
        void clear()
        {
            pin_src.reset();
            active = enqueued = false;
            src = nullptr;
            n = nh = 0;
            stages = 0;
        }
    } pending;

    int ensure_pipeline(size_t n_events)
    {
        if (!s_in)
            SZ_HIP(hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking));
        if (!s_out)
            SZ_HIP(hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking));
        if (!pin_tot)
            SZ_HIP(hipHostMalloc(reinterpret_cast<void **>(&pin_tot), kPinTot * sizeof(uint64_t), hipHostMallocDefault));
        while (ev.size() < n_events) {
            hipEvent_t e = nullptr;
            SZ_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ev.push_back(e);
        }
        return SZ_OK;
    }
    void sync_all()  // every path out of a pipelined call, errors included: nothing of it is in flight any more
    {
        if (s_in)
            (void)hipStreamSynchronize(s_in);
        if (stream)
            (void)hipStreamSynchronize(stream);
        if (s_out)
            (void)hipStreamSynchronize(s_out);
    }

    int ensure_stream()
    {
        int dev = 0;
        SZ_HIP(hipGetDevice(&dev));
        if (stream && dev != device)
            release();
        if (!stream) {
            SZ_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
            device = dev;
        }
        return SZ_OK;
    }
    int reserve(int i, size_t bytes)
    {
        if (bytes <= cap[i])
            return SZ_OK;
        if (buf[i]) {
            (void)hipFree(buf[i]);
            buf[i] = nullptr;
            cap[i] = 0;
        }
        size_t want = bytes + bytes / 4 + 4096;
        SZ_HIP(hipMalloc(&buf[i], want));
        cap[i] = want;
        return SZ_OK;
    }
    void release()
    {
        for (int i = 0; i < 5; i++) {
            if (buf[i])
                (void)hipFree(buf[i]);
            buf[i] = nullptr;
            cap[i] = 0;
        }
        sync_all();
        pending.clear();  // (a call begun and never finished: its work is drained, the caller's memory let go)
        for (hipEvent_t e : ev)
            (void)hipEventDestroy(e);
        ev.clear();
        if (pin_tot)
            (void)hipHostFree(pin_tot);
        pin_tot = nullptr;
        if (s_in)
            (void)hipStreamDestroy(s_in);
        if (s_out)
            (void)hipStreamDestroy(s_out);
        s_in = s_out = nullptr;
        if (stream) {
            drop_dual_scratch(device, stream);
            (void)hipStreamDestroy(stream);
        }
        stream = nullptr;
        device = -1;
    }
};
// Deliberately not freed at thread exit (HIP may be gone by then): a host thread that used the
// host-buffer API or the RoCC surface calls sz_release_thread_resources() before it ends (snappy_hip.h).
thread_local Workspace tl_ws;

// The stages of the pipeline, in input bytes (multiples of 64 KiB but the last).  The first copy in and the last
// compress + copy out are the parts of a call that nothing overlaps, so the stages start small (64 MiB), double up to
// 256 MiB -- a compressible stage of 4 096 fragments is a launch the persistent kernel takes at 63 GB/s, above the
// link's 50; one of 1 024 fragments gets 36, one of 512 only 20 -- and come down again towards the end: a stage takes
// at most a third of what is left.  2 GiB: 64 64 128 256 x 4 128 128 64 x 4 MiB.
constexpr size_t kPipelineMinBytes = (size_t)128 << 20;
constexpr size_t kStageMin = (size_t)64 << 20, kStageMax = (size_t)256 << 20;
std::vector<size_t> pipeline_stages(size_t n)
{
    std::vector<size_t> st;
    if (n < kPipelineMinBytes) {
        st.push_back(n);
        return st;
    }
    size_t rem = n, ramp = kStageMin;
    while (rem) {
        size_t s = kStageMin;
        while (s * 2 <= rem / 3 && s * 2 <= kStageMax && s * 2 <= ramp)
            s *= 2;
        if (rem <= s + kStageMin / 2)
            s = rem;
        st.push_back(s);
        rem -= s;
        if (st.size() >= 2 && ramp < kStageMax)
            ramp *= 2;
    }
    return st;
}

size_t put_varint32(uint8_t *dst, uint32_t v)
{
    size_t k = 0;
    while (v >= 0x80) {
        dst[k++] = (uint8_t)(v | 0x80);
        v >>= 7;
    }
    dst[k++] = (uint8_t)v;
    return k;
}

#ifdef SZ_HOST_TIMING  // diagnostic builds: where a pipelined call's time goes, to stderr
#include <chrono>
#define HOST_T(v) const double v = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count()
#else
#define HOST_T(v) ((void)0)
#endif

struct SyncOnExit {  // declared after the HostPins of a call: runs before they release the caller's memory
    Workspace &ws;
    ~SyncOnExit() { ws.sync_all(); }
};

}  // namespace

extern "C" {

size_t sz_max_compressed_length(size_t n) { return 32 + n + n / 6; }

int sz_shard_range(uint64_t n_blocks_total, uint32_t rank, uint32_t world, uint64_t *begin, uint64_t *end)
{
    if (world == 0 || rank >= world || !begin || !end)
        return SZ_EINVAL;
    const uint64_t base = n_blocks_total / world, rem = n_blocks_total % world;
    const uint64_t b = (uint64_t)rank * base + (rank < rem ? rank : rem);
    *begin = b;
    *end = b + base + (rank < rem ? 1u : 0u);
    return SZ_OK;
}

int sz_stream_offsets(const uint32_t *lens, uint64_t n_blocks, uint64_t base, uint64_t slot_stride,
                      uint64_t *offsets, uint64_t *bad_block)
{
    if (!offsets || (n_blocks && !lens))
        return SZ_EINVAL;
    uint64_t o = base;
    for (uint64_t i = 0; i < n_blocks; i++) {
        offsets[i] = o;
        if (lens[i] >= 0xfffffffdu || (slot_stride && lens[i] > slot_stride)) {
            if (bad_block)
                *bad_block = i;
            return SZ_ECORRUPT;
        }
        o += lens[i];
    }
    offsets[n_blocks] = o;
    return SZ_OK;
}

int sz_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        tl_hip_error = (int)e;
        return 0;
    }
    return n;
}

int sz_last_hip_error(void) { return tl_hip_error; }

// (the compiler the kernels were built with is part of the version: their speed -- and once their liveness -- has
// depended on how one clang merged branches and allocated registers; csrc/Makefile stamps "untuned toolchain" when it
// is not the one the numbers under profiles/ were measured with.  `src <id>` is the identity of the sources this
// library was built from -- csrc/Makefile SOURCE_ID, recomputed by tests/build_util.py source_id() --: the GPU box
// loads whatever .so was pushed, the GPU suite's first test refuses one that was not built from the tree beside it.)
#define SZ_STR2(x) #x
#define SZ_STR(x) SZ_STR2(x)
#ifndef SZ_TOOLCHAIN
#define SZ_TOOLCHAIN "HIP unknown, untuned toolchain"
#endif
#ifndef SZ_SOURCE_ID
#define SZ_SOURCE_ID "unknown"
#endif
const char *sz_version(void)
{
    return "snappy_hip 0.5 (gfx950; Snappy 1.1.8 bit-exact; built with clang " SZ_STR(__clang_major__) "." SZ_STR(__clang_minor__) "." SZ_STR(__clang_patchlevel__)
           ", " SZ_TOOLCHAIN "; src " SZ_SOURCE_ID ")";
}

const char *sz_strerror(int code)
{
    switch (code) {
    case SZ_OK: return "ok";
    case SZ_EINVAL: return "invalid argument";
    case SZ_ENOSPC: return "destination too small";
    case SZ_EHIP: return "HIP runtime error";
    case SZ_ENODEV: return "no usable GPU";
    case SZ_ECORRUPT: return "malformed Snappy stream";
    default: return "unknown error";
    }
}

void sz_release_thread_resources(void) { tl_ws.release(); }

int sz_release_stream_resources(void *stream)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess)
        return SZ_ENODEV;
    hipStream_t s = static_cast<hipStream_t>(stream);
    SZ_HIP(hipStreamSynchronize(s));  // the workspace may still be in use by launches on this stream
    drop_dual_scratch(dev, s);
    return SZ_OK;
}

/* ------------------------------------------------------------------ device-resident API */

namespace {
struct ProfileHookOneShot {  // sz_profile_next_launch arms the hook for exactly the call that follows it
    ~ProfileHookOneShot() { szk::drop_profile_events(); }
};
}  // namespace

int sz_compress_blocks_hip(const void *d_src, size_t in_stride, const uint32_t *d_in_len,
                           uint32_t n_blocks, void *d_dst, size_t dst_stride, uint32_t *d_out_len,
                           uint32_t flags, void *stream)
{
    ProfileHookOneShot one_shot;
    if (n_blocks == 0)
        return SZ_OK;
    if (!d_src || !d_dst || !d_out_len || (flags & ~(SZ_FLAG_BLOCK_VARINT | SZ_FLAG_HW_COMPAT)))
        return SZ_EINVAL;
    if (in_stride == 0 || (!d_in_len && in_stride > SZ_BLOCK_SIZE))
        return SZ_EINVAL;
    // with explicit lengths the host cannot see them; a block never exceeds min(in_stride, 64 KiB)
    const size_t longest = in_stride < SZ_BLOCK_SIZE ? in_stride : SZ_BLOCK_SIZE;
    if (dst_stride < sz_max_compressed_length(longest) + 5)
        return SZ_ENOSPC;
    if (flags & SZ_FLAG_HW_COMPAT) {  // the accelerator's own element stream (its worst case, n + n / 60 + 1, fits the slot too)
        SZ_HIP(szk::launch_compress_hw_blocks(d_src, in_stride, d_in_len, 0, n_blocks, d_dst, dst_stride, d_out_len,
                                              flags, static_cast<hipStream_t>(stream)));
        return SZ_OK;
    }
#ifdef SZ_ABLATION  // diagnostic builds only: timing ablations that produce wrong output
    if (const char *dbg = getenv("SZ_DEBUG_FLAGS"))
        flags |= (uint32_t)strtoul(dbg, nullptr, 0) & 0x7ff00u;
#endif
    szk::DualScratch view;
    std::shared_ptr<DualSlot> slot = dual_slot_for(static_cast<hipStream_t>(stream), n_blocks, in_stride, d_in_len, &view);
    std::unique_lock<std::mutex> seq = lock_live_slot(slot);  // one launch sequence at a time per workspace
    SZ_HIP(szk::launch_compress_blocks(d_src, in_stride, d_in_len, n_blocks, d_dst, dst_stride, d_out_len,
                                       flags, slot ? &view : nullptr, static_cast<hipStream_t>(stream)));
    return SZ_OK;
}

int sz_profile_next_launch(void *single_start, void *single_stop, void *multi_start, void *multi_stop)
{
    szk::set_profile_events(static_cast<hipEvent_t>(single_start), static_cast<hipEvent_t>(single_stop),
                            static_cast<hipEvent_t>(multi_start), static_cast<hipEvent_t>(multi_stop));
    return SZ_OK;
}

int sz_compress_stream_hip(const void *d_src, size_t n, void *d_slots, size_t slot_stride,
                           uint32_t *d_len, uint32_t flags, void *stream)
{
    ProfileHookOneShot one_shot;
    if (n == 0)
        return SZ_OK;
    if (!d_src || !d_slots || !d_len || n > 0xffffffffull || (flags & ~SZ_FLAG_BLOCK_VARINT))
        return SZ_EINVAL;
    const size_t longest = n < SZ_BLOCK_SIZE ? n : SZ_BLOCK_SIZE;
    if (slot_stride < sz_max_compressed_length(longest) + 5)
        return SZ_ENOSPC;
    const uint32_t n_blocks = (uint32_t)((n + SZ_BLOCK_SIZE - 1) / SZ_BLOCK_SIZE);
    szk::DualScratch view;
    std::shared_ptr<DualSlot> slot = dual_slot_for(static_cast<hipStream_t>(stream), n_blocks, SZ_BLOCK_SIZE, nullptr, &view);
    std::unique_lock<std::mutex> seq = lock_live_slot(slot);  // one launch sequence at a time per workspace
    SZ_HIP(szk::launch_compress_stream(d_src, n, d_slots, slot_stride, d_len, flags, slot ? &view : nullptr,
                                       static_cast<hipStream_t>(stream)));
    return SZ_OK;
}

int sz_pack_stream_hip(const void *d_slots, size_t slot_stride, const uint32_t *d_len,
                       uint32_t n_blocks, uint64_t base, void *d_packed, uint64_t *d_off, void *stream)
{
    if (!d_off || (n_blocks && (!d_slots || !d_len || !d_packed)))
        return SZ_EINVAL;
    SZ_HIP(szk::launch_pack_stream(d_slots, slot_stride, d_len, n_blocks, base, d_packed, d_off,
                                   static_cast<hipStream_t>(stream)));
    return SZ_OK;
}

int sz_uncompress_blocks_hip(const void *d_src, size_t src_stride, const uint32_t *d_src_len,
                             uint32_t n_streams, void *d_dst, size_t dst_stride, uint32_t *d_out_len,
                             int32_t *d_status, void *stream)
{
    if (n_streams == 0)
        return SZ_OK;
    if (!d_src || !d_src_len || !d_dst || !d_out_len || !d_status)
        return SZ_EINVAL;
    SZ_HIP(szk::launch_uncompress_streams(d_src, src_stride, d_src_len, n_streams, d_dst, dst_stride,
                                          d_out_len, d_status, static_cast<hipStream_t>(stream)));
    return SZ_OK;
}

int sz_uncompress_packed_hip(const void *d_packed, const uint64_t *d_off, uint32_t n_blocks,
                             uint64_t total_ulen, void *d_dst, uint32_t *d_out_len, int32_t *d_status,
                             void *stream)
{
    if (n_blocks == 0)
        return total_ulen == 0 ? SZ_OK : SZ_EINVAL;
    if (!d_packed || !d_off || !d_dst || !d_out_len || !d_status)
        return SZ_EINVAL;
    if (total_ulen == 0 || (total_ulen + SZ_BLOCK_SIZE - 1) / SZ_BLOCK_SIZE != n_blocks)
        return SZ_EINVAL;  // n_blocks must be ceil(total_ulen / 64 KiB)
    SZ_HIP(szk::launch_uncompress_packed(d_packed, d_off, n_blocks, total_ulen, d_dst, d_out_len, d_status,
                                         static_cast<hipStream_t>(stream)));
    return SZ_OK;
}

size_t sz_uncompress_stream_work_bytes(size_t n, uint64_t ulen)
{
    const size_t n_frag = (size_t)((ulen + SZ_BLOCK_SIZE - 1) / SZ_BLOCK_SIZE);
    return ((szk::stream_index_scratch_bytes(n) + 15) & ~(size_t)15) + (n_frag + 1) * sizeof(uint64_t) + n_frag * 8 + 64;
}

int sz_uncompress_stream_hip(const void *d_src, size_t n, size_t body, uint64_t ulen, void *d_dst, void *d_work,
                             size_t work_bytes, int32_t *d_status3, void *stream)
{
    if (!d_src || !d_dst || !d_work || !d_status3 || n == 0 || body == 0 || body > 5 || body > n ||
        n > SZ_STREAM_INDEX_MAX_BYTES || ulen == 0 || ulen > 0xffffffffull)
        return SZ_EINVAL;
    if (work_bytes < sz_uncompress_stream_work_bytes(n, ulen))
        return SZ_ENOSPC;
    const size_t n_frag = (size_t)((ulen + SZ_BLOCK_SIZE - 1) / SZ_BLOCK_SIZE);
    uint8_t *w = static_cast<uint8_t *>(d_work);
    const size_t idx_bytes = (szk::stream_index_scratch_bytes(n) + 15) & ~(size_t)15;
    uint64_t *d_off = reinterpret_cast<uint64_t *>(w + idx_bytes);
    uint32_t *d_flen = reinterpret_cast<uint32_t *>(d_off + n_frag + 1);
    int32_t *d_fst = reinterpret_cast<int32_t *>(d_flen + n_frag);
    hipStream_t s = static_cast<hipStream_t>(stream);
    uint32_t *d_res = reinterpret_cast<uint32_t *>(d_status3);
    SZ_HIP(szk::launch_stream_index(d_src, n, (uint32_t)body, ulen, w, d_off, d_res, s));
    SZ_HIP(szk::launch_uncompress_packed(d_src, d_off, (uint32_t)n_frag, ulen, d_dst, d_flen, d_fst, s));
    SZ_HIP(szk::launch_count_failed(d_fst, (uint32_t)n_frag, d_res, s));
    return SZ_OK;
}

/* --------------------------------------------------------------------- host-buffer API */

int sz_get_uncompressed_length(const void *src, size_t n, size_t *result)
{
    if (!src || !result)
        return SZ_EINVAL;
    const uint8_t *p = static_cast<const uint8_t *>(src);
    uint32_t v = 0;
    for (size_t i = 0; i < 5 && i < n; i++) {
        uint32_t b = p[i];
        if (i == 4 && b > 0x0f)
            return SZ_ECORRUPT;
        v |= (b & 0x7f) << (7 * i);
        if (b < 0x80) {
            *result = v;
            return SZ_OK;
        }
    }
    return SZ_ECORRUPT;
}

// One stage for inputs below 128 MiB, otherwise a pipeline of 64..256 MiB stages:
//   s_in:    copy in of stage k+1 ...
//   stream:  ... under fragments of stage k compressed into slots, packed behind stage k-1's bytes (the base comes
//            from device memory: no host in between), stage k's running total copied to pinned memory ...
//   s_out:   ... under the copy out of stage k-1, straight to its final place in `dst`.
// sz_compress_begin enqueues everything that does not need `dst` (0.3 ms of host time for 2 GiB); sz_compress_finish
// waits for one event per stage (it needs the total to know what to copy out and whether it fits), issues the copies
// out and synchronises everything once, at the end.  (Round 5: one copy in, kernels, a synchronisation, one copy out, a
// synchronisation, all on pageable memory: 15 GB/s on incompressible data, 9 on text, PCIe included.)
int sz_compress_begin(const void *src, size_t n)
{
    if ((n && !src) || n > 0xffffffffull)
        return SZ_EINVAL;
    if (sz_device_count() < 1)  // fail loudly: there is no CPU fallback
        return SZ_ENODEV;
    Workspace &ws = tl_ws;
    Workspace::Pending &p = ws.pending;
    if (p.active)
        return SZ_EINVAL;  // one call in flight per host thread (one RoCC port per hart, cmd.ready := !busy)
    int rc = ws.ensure_stream();
    if (rc)
        return rc;
    p.clear();
    p.src = static_cast<const uint8_t *>(src);
    p.n = n;
    p.nh = put_varint32(p.hdr, (uint32_t)n);
    p.active = true;
    if (n <= SZ_BLOCK_SIZE)  // nothing or one fragment: all of it in sz_compress_finish, behind one synchronisation
        return SZ_OK;
    struct Undo {  // an error on the way: nothing of the call stays in flight, the caller's memory is let go
        Workspace &ws;
        bool armed = true;
        ~Undo()
        {
            if (armed) {
                ws.sync_all();
                ws.pending.clear();
            }
        }
    } undo{ws};
    HOST_T(t_0);
    const uint32_t nb = (uint32_t)((n + SZ_BLOCK_SIZE - 1) / SZ_BLOCK_SIZE);
    const size_t stride = SZ_SLOT_STRIDE;
    const std::vector<size_t> stages = pipeline_stages(n);
    const uint32_t K = (uint32_t)stages.size();
    size_t longest = 0;
    for (size_t b : stages)
        longest = b > longest ? b : longest;
    const uint32_t cb = (uint32_t)((longest + SZ_BLOCK_SIZE - 1) / SZ_BLOCK_SIZE);  // fragments of the longest stage
    if (K > Workspace::kPinTot)
        return SZ_EINVAL;  // (cannot happen: n < 2^32 gives fewer than 30 stages)
    if ((rc = ws.reserve(0, n)) || (rc = ws.reserve(1, (size_t)cb * stride)) ||
        (rc = ws.reserve(2, (size_t)nb * sizeof(uint32_t))) ||
        (rc = ws.reserve(3, ((size_t)nb + K) * sizeof(uint64_t))) ||
        (rc = ws.reserve(4, sz_max_compressed_length(n) + 16)) || (rc = ws.ensure_pipeline(2 * (size_t)K)))
        return rc;
    uint8_t *d_in = static_cast<uint8_t *>(ws.buf[0]);
    uint8_t *d_slots = static_cast<uint8_t *>(ws.buf[1]);
    uint32_t *d_len = static_cast<uint32_t *>(ws.buf[2]);
    uint64_t *d_off = static_cast<uint64_t *>(ws.buf[3]);  // stage k: its fragments' offsets + 1 entry (its running total), one stage behind the other
    uint8_t *d_packed = static_cast<uint8_t *>(ws.buf[4]);
    if (K > 1)
        p.pin_src.reset(new HostPin(src, n, true));
    HOST_T(t_1);
    size_t o = 0, first_frag = 0;
    const uint64_t *prev_total = nullptr;
    uint64_t *pin_tot_dev = nullptr;
    SZ_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&pin_tot_dev), ws.pin_tot, 0));
    for (uint32_t k = 0; k < K; k++) {
        const size_t bytes = stages[k];
        const uint32_t nbk = (uint32_t)((bytes + SZ_BLOCK_SIZE - 1) / SZ_BLOCK_SIZE);
        uint64_t *off_k = d_off + first_frag + k;
        SZ_HIP(hipMemcpyAsync(d_in + o, p.src + o, bytes, hipMemcpyHostToDevice, ws.s_in));
        SZ_HIP(hipEventRecord(ws.ev[2 * k], ws.s_in));
        SZ_HIP(hipStreamWaitEvent(ws.stream, ws.ev[2 * k], 0));
        if ((rc = sz_compress_stream_hip(d_in + o, bytes, d_slots, stride, d_len + first_frag, 0, ws.stream)))
            return rc;
        // (the stage's running total goes to pinned memory by a store of the scan kernel, not by a copy of 8 bytes: see there)
        SZ_HIP(szk::launch_pack_stream(d_slots, stride, d_len + first_frag, nbk, p.nh, d_packed, off_k, ws.stream, prev_total, pin_tot_dev + k));
        SZ_HIP(hipEventRecord(ws.ev[2 * k + 1], ws.stream));
        prev_total = off_k + nbk;
        o += bytes;
        first_frag += nbk;
    }
#ifdef SZ_HOST_TIMING
    HOST_T(t_2);
    fprintf(stderr, "sz_compress_begin %zu bytes, %u stages (src registered: %d): reserve + register %.2f ms, enqueue %.2f ms\n", n, K,
            (int)(p.pin_src && p.pin_src->ok), t_1 - t_0, t_2 - t_1);
#endif
    p.stages = K;
    p.enqueued = true;
    undo.armed = false;
    return SZ_OK;
}

int sz_compress_finish(void *dst, size_t dst_cap, size_t *out_len)
{
    Workspace &ws = tl_ws;
    Workspace::Pending &p = ws.pending;
    if (!p.active)
        return SZ_EINVAL;
    struct Done {  // every way out: nothing of the call in flight, the caller's memory let go, the thread free for the next call
        Workspace &ws;
        ~Done()
        {
            if (ws.pending.enqueued)
                ws.sync_all();
            ws.pending.clear();
        }
    } done_guard{ws};
    if (!dst && !out_len)
        return SZ_OK;  // the call is abandoned
    if (!dst || !out_len)
        return SZ_EINVAL;
    const size_t n = p.n, nh = p.nh;
    if (dst_cap < nh)
        return SZ_ENOSPC;
    memcpy(dst, p.hdr, nh);
    if (n == 0) {
        *out_len = nh;
        return SZ_OK;
    }
    int rc;
    if (n <= SZ_BLOCK_SIZE) {
        // One fragment (what the RoCC surface sends: one buffer per COMPRESS command, 130 bytes in tests/compression.c):
        // the block's own raw stream -- varint + body, SZ_FLAG_BLOCK_VARINT -- IS snappy::Compress's output, so there is
        // nothing to pack, and length and bytes come back in ONE copy behind ONE synchronisation (the length sits right
        // behind the slot): 67 -> ~40 us for 130 bytes against the general path (two more kernels, a copy and a
        // synchronisation for the length alone).
        const size_t bound = (sz_max_compressed_length(n) + 5 + 15) & ~(size_t)15;  // slot, and the offset of the length behind it
        if ((rc = ws.reserve(0, n)) || (rc = ws.reserve(1, bound + 16)))
            return rc;
        uint8_t *d_in = static_cast<uint8_t *>(ws.buf[0]);
        uint8_t *d_slot = static_cast<uint8_t *>(ws.buf[1]);
        ws.host.resize(bound + 16);
        SZ_HIP(hipMemcpyAsync(d_in, p.src, n, hipMemcpyHostToDevice, ws.stream));
        if ((rc = sz_compress_blocks_hip(d_in, n, nullptr, 1, d_slot, bound, reinterpret_cast<uint32_t *>(d_slot + bound),
                                         SZ_FLAG_BLOCK_VARINT, ws.stream)))
            return rc;
        SZ_HIP(hipMemcpyAsync(ws.host.data(), d_slot, bound + 4, hipMemcpyDeviceToHost, ws.stream));
        SZ_HIP(hipStreamSynchronize(ws.stream));
        uint32_t len = 0;
        memcpy(&len, ws.host.data() + bound, sizeof(len));
        if (len > bound)  // an error marker instead of a length
            return SZ_EHIP;
        if (len > dst_cap)
            return SZ_ENOSPC;
        memcpy(dst, ws.host.data(), len);  // (the varint it starts with is the one written above)
        *out_len = len;
        return SZ_OK;
    }
    HOST_T(t_0);
    const uint32_t K = p.stages;
    uint8_t *d_packed = static_cast<uint8_t *>(ws.buf[4]);
    uint8_t *h_dst = static_cast<uint8_t *>(dst);
    const size_t bound = sz_max_compressed_length(n);
    HostPin pin_dst(K > 1 ? dst : nullptr, dst_cap < bound ? dst_cap : bound, false);
    SyncOnExit drain{ws};  // (declared after the pin: nothing is in flight any more when `dst` is let go)
    HOST_T(t_1);
    uint64_t done = nh;  // bytes of the stream that are in `dst`, or on their way there
    // The copy out of a stage that did not compress is a KERNEL's (when `dst` is registered, i.e. device-visible): the copy
    // in keeps the DMA engines to itself.  With both directions on the engines the runtime put them on one engine now and
    // then (in bench.py's process: in every call on incompressible data) and a call took copy in + copy out instead of the
    // longer of the two (EXPERIMENTS.md R6.9).  A stage that compressed keeps the engine: its kernel (the team kernel: four
    // waves of 128 registers per SIMD) leaves a copy kernel no register to run beside it, and its copy out is short.
    // SZ_COPY_OUT=engine / kernel: one way for every stage.
    uint8_t *dst_dev = nullptr;
    int copy_mode = 0;  // 0: by stage, 1: engine, 2: kernel
    if (pin_dst.ok) {
        const char *e = getenv("SZ_COPY_OUT");
        copy_mode = !e ? 0 : strcmp(e, "engine") == 0 ? 1 : strcmp(e, "kernel") == 0 ? 2 : 0;
        if (copy_mode != 1 && hipHostGetDevicePointer(reinterpret_cast<void **>(&dst_dev), dst, 0) != hipSuccess) {
            (void)hipGetLastError();
            dst_dev = nullptr;
        }
    }
    const std::vector<size_t> stages = pipeline_stages(n);  // (the same cut as sz_compress_begin's)
    for (uint32_t k = 0; k < K; k++) {
        SZ_HIP(hipEventSynchronize(ws.ev[2 * k + 1]));
        const uint64_t total = *const_cast<volatile uint64_t *>(ws.pin_tot + k);
        if (total == ~0ull)  // a fragment came back with an error marker instead of a length
            return SZ_EHIP;
        if (total > dst_cap)
            return SZ_ENOSPC;
        const bool by_kernel = dst_dev && (copy_mode == 2 || (k < stages.size() && (total - done) * 8 >= (uint64_t)stages[k] * 7));
        if (by_kernel)
            SZ_HIP(szk::launch_copy_out(dst_dev + done, d_packed + done, total - done, ws.s_out));
        else
            SZ_HIP(hipMemcpyAsync(h_dst + done, d_packed + done, total - done, hipMemcpyDeviceToHost, ws.s_out));
        done = total;
    }
    SZ_HIP(hipStreamSynchronize(ws.s_out));
#ifdef SZ_HOST_TIMING
    HOST_T(t_2);
    fprintf(stderr, "sz_compress_finish (dst registered: %d): register %.2f ms, drain %.2f ms\n", (int)pin_dst.ok, t_1 - t_0, t_2 - t_1);
#endif
    *out_len = (size_t)done;
    return SZ_OK;
}

int sz_compress(const void *src, size_t n, void *dst, size_t dst_cap, size_t *out_len)
{
    if (!dst || !out_len || (n && !src) || n > 0xffffffffull)
        return SZ_EINVAL;
    uint8_t hdr[5];
    if (dst_cap < put_varint32(hdr, (uint32_t)n))
        return SZ_ENOSPC;
    const int rc = sz_compress_begin(src, n);
    return rc ? rc : sz_compress_finish(dst, dst_cap, out_len);
}

int sz_uncompress(const void *src, size_t n, void *dst, size_t dst_cap, size_t *out_len)
{
    if (!src || !out_len || n == 0 || n > 0xffffffffull)
        return n == 0 ? SZ_ECORRUPT : SZ_EINVAL;
    size_t ulen = 0;
    int rc = sz_get_uncompressed_length(src, n, &ulen);
    if (rc)
        return rc;
    if (ulen > dst_cap || (ulen && !dst))
        return SZ_ENOSPC;
    if (sz_device_count() < 1)
        return SZ_ENODEV;
    Workspace &ws = tl_ws;
    if ((rc = ws.ensure_stream()))
        return rc;
    if ((rc = ws.reserve(0, n + 8)) || (rc = ws.reserve(1, ulen + 16)) || (rc = ws.reserve(2, 16)))  // (the stream index reads up to n + 8)
        return rc;
    uint8_t *d_in = static_cast<uint8_t *>(ws.buf[0]);
    uint8_t *d_out = static_cast<uint8_t *>(ws.buf[1]);
    uint32_t *d_meta = static_cast<uint32_t *>(ws.buf[2]);  // [0]=src_len [1]=out_len [2]=status
    // (large streams: the caller's buffers registered for DMA for the length of the call, as in sz_compress)
    const bool large = ulen >= kPipelineMinBytes;
    HostPin pin_src(large ? src : nullptr, n, true), pin_dst(large ? dst : nullptr, ulen, false);
    SyncOnExit drain{ws};
    SZ_HIP(hipMemcpyAsync(d_in, src, n, hipMemcpyHostToDevice, ws.stream));

    // A stream of more than a few fragments: find the fragments' compressed offsets in parallel
    // (snappy_stream_index.hip) and decode one wavefront per fragment.  Anything unusual -- a stream
    // that was not produced in 64 KiB fragments, a malformed element -- falls through to the serial
    // decoder below, which also produces the error code.
    const size_t n_frag = (ulen + SZ_BLOCK_SIZE - 1) / SZ_BLOCK_SIZE;
    if (n_frag >= 4 && n <= SZ_STREAM_INDEX_MAX_BYTES && !getenv("SZ_SERIAL_UNCOMPRESS")) {
        size_t body = 0;
        while (body < 5 && (static_cast<const uint8_t *>(src)[body] & 0x80))
            body++;
        body++;
        const size_t work = sz_uncompress_stream_work_bytes(n, ulen);
        if ((rc = ws.reserve(3, work)))
            return rc;
        int32_t *d_st = reinterpret_cast<int32_t *>(d_meta);  // 3 words
        if (!large) {
            if ((rc = sz_uncompress_stream_hip(d_in, n, body, ulen, d_out, ws.buf[3], work, d_st, ws.stream)))
                return rc;
            int32_t st3[3] = {1, 1, 1};
            SZ_HIP(hipMemcpyAsync(st3, d_st, sizeof(st3), hipMemcpyDeviceToHost, ws.stream));
            SZ_HIP(hipStreamSynchronize(ws.stream));
            const bool good = st3[0] == 0 && st3[1] == 0 && st3[2] == 0;
            if (good) {
                SZ_HIP(hipMemcpyAsync(dst, d_out, ulen, hipMemcpyDeviceToHost, ws.stream));
                SZ_HIP(hipStreamSynchronize(ws.stream));
                *out_len = ulen;
                return SZ_OK;
            }
        } else {
            // The whole stream has to be on the device before its fragments can be found (the index reads all of it);
            // behind the index the fragments are decoded in stages of 4 096 and every stage's 256 MiB leave for `dst`
            // (s_out) under the decoding of the next one.  `dst` may have been written in part when the stream turns
            // out to be bad -- as with snappy::RawUncompress, which writes while it parses.
            const size_t stage_frags = 4096;
            const size_t K = (n_frag + stage_frags - 1) / stage_frags;
            if ((rc = ws.ensure_pipeline(K + 2)))
                return rc;
            uint8_t *w = static_cast<uint8_t *>(ws.buf[3]);
            const size_t idx_bytes = (szk::stream_index_scratch_bytes(n) + 15) & ~(size_t)15;  // (the layout of sz_uncompress_stream_hip's work area)
            uint64_t *d_off = reinterpret_cast<uint64_t *>(w + idx_bytes);
            uint32_t *d_flen = reinterpret_cast<uint32_t *>(d_off + n_frag + 1);
            int32_t *d_fst = reinterpret_cast<int32_t *>(d_flen + n_frag);
            uint32_t *d_res = reinterpret_cast<uint32_t *>(d_st);
            volatile int32_t *h_st = reinterpret_cast<volatile int32_t *>(ws.pin_tot);  // pinned: [0..2] after the index, [4..6] at the end
            SZ_HIP(szk::launch_stream_index(d_in, n, (uint32_t)body, ulen, w, d_off, d_res, ws.stream));
            SZ_HIP(hipMemcpyAsync(const_cast<int32_t *>(h_st), d_st, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, ws.stream));
            SZ_HIP(hipEventRecord(ws.ev[K], ws.stream));
            SZ_HIP(hipEventSynchronize(ws.ev[K]));
            if (h_st[0] == 0 && h_st[1] == 0) {
                for (size_t k = 0; k < K; k++) {
                    const size_t f0 = k * stage_frags, nf = n_frag - f0 < stage_frags ? n_frag - f0 : stage_frags;
                    const uint64_t sub = ulen - f0 * SZ_BLOCK_SIZE < nf * (uint64_t)SZ_BLOCK_SIZE ? ulen - f0 * SZ_BLOCK_SIZE : nf * (uint64_t)SZ_BLOCK_SIZE;
                    SZ_HIP(szk::launch_uncompress_packed(d_in, d_off + f0, (uint32_t)nf, sub, d_out + f0 * SZ_BLOCK_SIZE, d_flen + f0,
                                                         d_fst + f0, ws.stream));
                    SZ_HIP(hipEventRecord(ws.ev[k], ws.stream));
                }
                SZ_HIP(szk::launch_count_failed(d_fst, (uint32_t)n_frag, d_res, ws.stream));
                SZ_HIP(hipMemcpyAsync(const_cast<int32_t *>(h_st) + 4, d_st, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, ws.stream));
                SZ_HIP(hipEventRecord(ws.ev[K + 1], ws.stream));
                for (size_t k = 0; k < K; k++) {
                    const size_t o = k * stage_frags * SZ_BLOCK_SIZE, bytes = ulen - o < stage_frags * SZ_BLOCK_SIZE ? ulen - o : stage_frags * SZ_BLOCK_SIZE;
                    SZ_HIP(hipStreamWaitEvent(ws.s_out, ws.ev[k], 0));
                    SZ_HIP(hipMemcpyAsync(static_cast<uint8_t *>(dst) + o, d_out + o, bytes, hipMemcpyDeviceToHost, ws.s_out));
                }
                SZ_HIP(hipEventSynchronize(ws.ev[K + 1]));
                SZ_HIP(hipStreamSynchronize(ws.s_out));
                if (h_st[4] == 0 && h_st[5] == 0 && h_st[6] == 0) {
                    *out_len = ulen;
                    return SZ_OK;
                }
            }
        }
    }
    uint32_t meta[3] = {(uint32_t)n, 0, 0};
    SZ_HIP(hipMemcpyAsync(d_meta, meta, sizeof(meta), hipMemcpyHostToDevice, ws.stream));
    if (ulen <= SZ_BLOCK_SIZE) {
        // One fragment (the RoCC surface's UNCOMPRESS of what one COMPRESS wrote): decoded size, status and bytes come
        // back in ONE copy behind ONE synchronisation -- the two words sit right behind the output --, and `dst` is
        // written only when the stream was good.
        const size_t tail = (ulen + 15) & ~(size_t)15;
        if ((rc = ws.reserve(1, tail + 16)))
            return rc;
        d_out = static_cast<uint8_t *>(ws.buf[1]);
        uint32_t *d_res = reinterpret_cast<uint32_t *>(d_out + tail);  // [0] = decoded size, [1] = status
        if ((rc = sz_uncompress_blocks_hip(d_in, n, d_meta, 1, d_out, ulen, d_res, reinterpret_cast<int32_t *>(d_res + 1),
                                           ws.stream)))
            return rc;
        ws.host.resize(tail + 8);
        SZ_HIP(hipMemcpyAsync(ws.host.data(), d_out, tail + 8, hipMemcpyDeviceToHost, ws.stream));
        SZ_HIP(hipStreamSynchronize(ws.stream));
        uint32_t res[2];
        memcpy(res, ws.host.data() + tail, sizeof(res));
        if ((int32_t)res[1] != 0)
            return (int32_t)res[1];
        if (res[0] != ulen)
            return SZ_ECORRUPT;
        if (ulen)
            memcpy(dst, ws.host.data(), ulen);
        *out_len = ulen;
        return SZ_OK;
    }
    if ((rc = sz_uncompress_blocks_hip(d_in, n, d_meta, 1, d_out, ulen, d_meta + 1,
                                       reinterpret_cast<int32_t *>(d_meta + 2), ws.stream)))
        return rc;
    SZ_HIP(hipMemcpyAsync(meta, d_meta, sizeof(meta), hipMemcpyDeviceToHost, ws.stream));
    SZ_HIP(hipStreamSynchronize(ws.stream));
    if ((int32_t)meta[2] != 0)
        return (int32_t)meta[2];
    if (meta[1] != ulen)
        return SZ_ECORRUPT;
    if (ulen) {
        SZ_HIP(hipMemcpyAsync(dst, d_out, ulen, hipMemcpyDeviceToHost, ws.stream));
        SZ_HIP(hipStreamSynchronize(ws.stream));
    }
    *out_len = ulen;
    return SZ_OK;
}

/* ------------------------------------------------------------------ RoCC command surface */

int sz_rocc_cmd(int channel, int funct, uint64_t rs1, uint64_t rs2, uint64_t *rd)
{
    if (channel != SZ_ROCC_CHANNEL)  // the accelerator sits on custom3 only (Configs.scala:87-95)
        return SZ_EINVAL;
    switch (funct) {
    case SZ_ROCC_SET_LENGTH:  // CompressionAccelerator.scala:363-364: length := rs1 (32 bit)
        tl_rocc_length = (uint32_t)rs1;
        if (rd)
            *rd = 0;
        return SZ_OK;
    case SZ_ROCC_COMPRESS: {  // :365-384 + :405-407: rd := compressed length
        size_t out = 0;
        const size_t len = tl_rocc_length;
        int rc = sz_compress(reinterpret_cast<const void *>(rs1), len, reinterpret_cast<void *>(rs2),
                             sz_max_compressed_length(len), &out);
        if (rc)
            return rc;
        if (rd)
            *rd = out;
        return SZ_OK;
    }
    case SZ_ROCC_UNCOMPRESS: {  // tests/compression.c:32-39: rd != 0 means failure
        size_t ulen = 0, out = 0;
        const void *src = reinterpret_cast<const void *>(rs1);
        int rc = sz_get_uncompressed_length(src, tl_rocc_length, &ulen);
        if (rc == SZ_OK)
            rc = sz_uncompress(src, tl_rocc_length, reinterpret_cast<void *>(rs2), ulen, &out);
        if (rc == SZ_ENODEV || rc == SZ_EHIP)
            return rc;
        if (rd)
            *rd = rc == SZ_OK ? 0 : 1;
        return SZ_OK;
    }
    default:
        return SZ_EINVAL;
    }
}

uint64_t rocc_cmd(int channel, int funct, uint64_t rs1, uint64_t rs2)
{
    uint64_t rd = 0;
    int rc = sz_rocc_cmd(channel, funct, rs1, rs2, &rd);
    tl_rocc_error = rc;
    if (rc != SZ_OK) {
        fprintf(stderr, "snappy_hip: RoCC command funct=%d failed: %s (hip error %d)\n", funct,
                sz_strerror(rc), sz_last_hip_error());
        return (uint64_t)-1;
    }
    return rd;
}

int sz_rocc_last_error(void) { return tl_rocc_error; }

}  // extern "C"
