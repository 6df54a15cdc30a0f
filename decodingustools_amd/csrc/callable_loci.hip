// callable_loci.hip -- implementation of include/callable_loci.h for MI355X (gfx950).
//
// Host side of the engine: device buffers, uploads, kernel launches, event timing.  The work
// itself is in kernels.hip.h.  No CPU fallback exists: without a HIP device cl_create fails.
#include "../../include/callable_loci.h"
#include "kernels.hip.h"
#include "host_parallel.h"
#include "qual_pack.h"
#include "pass_rows.h"

#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

using namespace clk;

namespace {

// rocTX ranges around the host-visible phases (rocprofv3 --marker-trace shows them); bound at run time so
// that the library does not depend on the profiler's marker library being installed -- and only when that library
// is in the process already (a profiler brought it) or DUT_ROCTX=1 asks for it: loading it cold took 35 ms of a
// process's first contig
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        const char *want = getenv("DUT_ROCTX");
        const int mode = RTLD_NOW | RTLD_LOCAL | ((want && *want == '1') ? 0 : RTLD_NOLOAD);
        for (const char *lib : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
            void *h = dlopen(lib, mode);
            if (!h) continue;
            push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
            pop = (int (*)())dlsym(h, "roctxRangePop");
            if (push && pop) return;
            push = nullptr; pop = nullptr;
        }
    }
};
struct Range {
    static const Roctx &rt() { static const Roctx r; return r; }
    explicit Range(const char *name) { if (rt().push) rt().push(name); }
    ~Range() { if (rt().pop) rt().pop(); }
};


// window size (reference positions per workgroup).  2048 -> about 19 KB of LDS per workgroup of the short-read
// variant, eight workgroups (32 waves) per CU.
#ifndef CL_WINDOW
#define CL_WINDOW 2048
#endif
constexpr uint32_t kT = CL_WINDOW;

// Host staging array of a trivially copyable type that grows without value-initialising what it adds (a contig's
// per-read arrays are hundreds of megabytes: zero-filling them before they are overwritten showed) and appends in
// parallel chunks.  Throws std::bad_alloc like a vector.
template <typename T> struct RawVec {
    T *p = nullptr;
    size_t n = 0, cap = 0;
    RawVec() = default;
    RawVec(const RawVec &) = delete;
    RawVec &operator=(const RawVec &) = delete;
    RawVec(RawVec &&o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = o.cap = 0; }
    ~RawVec() { free(p); }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T *data() { return p; }
    const T *data() const { return p; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    const T &back() const { return p[n - 1]; }
    void clear() { n = 0; }
    void release() { free(p); p = nullptr; n = cap = 0; }
    void swap(RawVec &o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(cap, o.cap); }
    void reserve(size_t want)
    {
        if (want <= cap) return;
        size_t nc = std::max(want, cap + cap / 2 + 16);
        T *q = static_cast<T *>(realloc(p, nc * sizeof(T)));
        if (!q) throw std::bad_alloc();
        p = q; cap = nc;
    }
    void resize(size_t m) { reserve(m); n = m; }                 // new elements are NOT initialised
    void push_back(const T &v) { reserve(n + 1); p[n++] = v; }
    void append(const T *src, size_t m)                          // parallel copy
    {
        reserve(n + m);
        T *dst = p + n;
        const size_t grain = (4u << 20) / sizeof(T);
        dut::parallel_for((m + grain - 1) / grain, 1, [&](size_t k) {
            const size_t a = k * grain, b = std::min(m, a + grain);
            memcpy(dst + a, src + a, (b - a) * sizeof(T));
        });
        n += m;
    }
};

// DUT_TIMING=1: wall-clock of the engine's host stages on stderr (tooling; off by default)
struct StageTimer {
    bool on;
    double t0;
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    StageTimer() : on(getenv("DUT_TIMING") && *getenv("DUT_TIMING") == '1'), t0(on ? now() : 0.0) {}
    void lap(const char *what)
    {
        if (!on) return;
        const double t1 = now();
        fprintf(stderr, "[dut-timing]     engine: %-24s %8.1f ms\n", what, (t1 - t0) * 1e3);
        t0 = t1;
    }
};

template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;     // elements
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = n + n / 8 + 64;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), want * sizeof(T));
        if (e != hipSuccess) { p = nullptr; return e; }
        cap = want;
        return hipSuccess;
    }
    // like reserve, but the first `used` elements survive a reallocation
    hipError_t grow_keep(size_t n, size_t used, hipStream_t stream)
    {
        if (n <= cap) return hipSuccess;
        if (!p || used == 0) return reserve(n);
        T *q = nullptr;
        size_t want = n + n / 4 + 64;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&q), want * sizeof(T));
        if (e != hipSuccess) return e;
        e = hipMemcpyAsync(q, p, used * sizeof(T), hipMemcpyDeviceToDevice, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) { (void)hipFree(q); return e; }
        (void)hipFree(p);
        p = q; cap = want;
        return hipSuccess;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

} // namespace

// Pinned staging ring for host-to-device copies, one per device and process (its contexts share it; a transfer holds
// it from start to finish): kCopyThreads host threads, each with its own stream and two pinned buffers; a thread fills
// one buffer (memcpy from the caller's pageable memory, or records built in place) while the DMA of its other buffer
// runs, so the link sees pinned memory only and the fills of all threads overlap all transfers.
struct PinRing {
    static constexpr int kCopyThreads = 16;                 // buffer pairs: plain copies use threads() of them, the walkers
                                                            // that produce a stream into the buffers (rows, run table) all
    static constexpr size_t kPinBytes = 4u << 20;
    static int threads()                                    // DUT_COPY_THREADS (1..16), default 8
    {
        static const int n = [] {
            const char *e = getenv("DUT_COPY_THREADS");
            const int v = e ? atoi(e) : 8;
            return v < 1 ? 1 : (v > kCopyThreads ? kCopyThreads : v);
        }();
        return n;
    }
    int device = 0;
    hipStream_t copy_stream[kCopyThreads] = {};
    uint8_t *pin[kCopyThreads][2] = {};
    hipEvent_t pin_ev[kCopyThreads][2] = {};
    // who is using the ring: a transfer holds it from ring_start to ring_finish -- across C-ABI calls for a quality
    // prefetch, and possibly released on another thread than the one that took it, so an ownership flag under a
    // condition variable rather than a mutex (unlocking a std::mutex from another thread is undefined)
    std::mutex own_mu;
    std::condition_variable own_cv;
    const void *owner = nullptr;
    void acquire(const void *who)
    {
        std::unique_lock<std::mutex> lk(own_mu);
        // a context of this device with an unclaimed prefetch pins the ring until its next push / upload / begin /
        // abort / destroy; a thread that drives two contexts must not interleave them there
        int waited = 0;
        while (owner && !own_cv.wait_for(lk, std::chrono::seconds(10), [this] { return owner == nullptr; }))
            if (++waited == 1)
                fprintf(stderr, "[callable_loci] waiting for the device's pinned staging ring: another context holds it "
                                "(an unclaimed cl_contig_prefetch_qual keeps it until that context's next push, upload, begin, abort or destroy)\n");
        owner = who;
    }
    void release(const void *who)
    {
        { std::lock_guard<std::mutex> g(own_mu); if (owner == who) owner = nullptr; }
        own_cv.notify_one();
    }
    // The ring's own threads: started once (at cl_create), asleep between transfers.  (A thread created per transfer had to
    // wait for the process's address-space lock whenever another thread was giving a few hundred megabytes back to the
    // system -- 26 ms in front of a 3 ms transfer, measured: profiles/r04_first_pass_stages.txt.)
    dut::Crew crew;
    bool ok = false;
    int slots = 0;                                          // thread slots that have their stream, buffers and events
    // slots [slots, n) are made (by the ring's owner, or at construction); false when the runtime refuses
    bool ensure_slots(int n)
    {
        if (n > kCopyThreads) n = kCopyThreads;
        if (hipSetDevice(device) != hipSuccess) return false;
        for (int t = slots; t < n; ++t) {
            if (hipStreamCreateWithFlags(&copy_stream[t], hipStreamNonBlocking) != hipSuccess) return false;
            for (int b = 0; b < 2; ++b) {
                if (hipHostMalloc(reinterpret_cast<void **>(&pin[t][b]), kPinBytes, hipHostMallocDefault) != hipSuccess) return false;
                // blocking waits: a copier that spins on its buffer's event burns a core the host stages beside it
                // need (DUT_PIN_SPIN=1: the runtime's default busy wait, for comparison)
                const char *spin = getenv("DUT_PIN_SPIN");
                const unsigned flags = hipEventDisableTiming | ((spin && *spin == '1') ? 0u : (unsigned)hipEventBlockingSync);
                if (hipEventCreateWithFlags(&pin_ev[t][b], flags) != hipSuccess) return false;
            }
            slots = t + 1;
        }
        return true;
    }
    explicit PinRing(int dev) : device(dev) { ok = ensure_slots(kCopyThreads); if (ok) crew.ensure(kCopyThreads); }
    ~PinRing()
    {
        (void)hipSetDevice(device);
        for (int t = 0; t < kCopyThreads; ++t) {
            for (int b = 0; b < 2; ++b) {
                if (pin_ev[t][b]) (void)hipEventDestroy(pin_ev[t][b]);
                if (pin[t][b]) (void)hipHostFree(pin[t][b]);
            }
            if (copy_stream[t]) (void)hipStreamDestroy(copy_stream[t]);
        }
    }
    PinRing(const PinRing &) = delete;
    PinRing &operator=(const PinRing &) = delete;
};

static std::shared_ptr<PinRing> acquire_ring(int device)
{
    static std::mutex mu;
    static std::map<int, std::weak_ptr<PinRing>> rings;
    std::lock_guard<std::mutex> g(mu);
    std::shared_ptr<PinRing> r = rings[device].lock();
    if (!r) {
        r = std::make_shared<PinRing>(device);
        if (!r->ok) return nullptr;
        rings[device] = r;
    }
    return r;
}

// The host staging arrays of a contig (hundreds of megabytes) are only needed between cl_contig_begin and
// cl_contig_upload.  A context hands them to this process-wide pool when its contig is uploaded and takes a set back
// at its next cl_contig_begin -- so does a fresh context: memory that has been touched before is written at several
// times the rate of newly mapped pages (staging a chr21-sized contig: 5 ms against 31 ms), and callers that keep one
// context per resident contig would otherwise fault a new set in for every contig.  At most kStagingSets sets are kept.
struct StagingSet {
    RawVec<uint8_t> ref, mapq;
    RawVec<int32_t> pos;
    RawVec<uint32_t> cigar_off, cigar, end, ck_x, ck_y, rec_cnt, rec_of;
    RawVec<unsigned long long> qual_off, rb_off;
    RawVec<uint64_t> qbits;
};
constexpr size_t kStagingSets = 2;
static std::mutex g_staging_mu;
static std::vector<std::unique_ptr<StagingSet>> g_staging;

// config 5: the resident tile of the site pileup (cl_site_upload) and the buffers of a run
struct SiteResident {
    DevBuf<SiteRec> rec; DevBuf<uint8_t> seq; DevBuf<uint32_t> cig, p0, ix, hist, bk; DevBuf<unsigned long long> base;
    uint64_t n = 0, ncig = 0, nbase = 0, ref_len = 0;
    uint32_t contig_len = 0;
    bool resident = false;
    bool filtered = false;           // the resident tile holds only the reads that overlap a site of the list it was uploaded for (cl_site_pileup)
    void release() { rec.release(); seq.release(); cig.release(); p0.release(); ix.release(); hist.release(); bk.release(); base.release(); resident = false; }
};

struct cl_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint64_t host_max_end = 0;         // largest pos + reference span over the pushed reads (32-bit clamped spans), for the extent
    std::shared_ptr<PinRing> ring;                    // the device's pinned staging ring (shared by its contexts)
    bool crew_busy = false;                           // a transfer in flight on the ring's threads (waited for by ring_finish)
    dut::Thread prealloc;                             // cl_contig_reserve: the device buffers of the contig being pushed, allocated beside the push
    hipError_t copy_err[PinRing::kCopyThreads] = {};
    bool ring_held = false;                           // this context holds the ring's lock (ring_start .. ring_finish)
    double ring_t0 = 0;                               // DUT_TIMING: when the transfer in flight was started
    // cl_contig_prefetch_qual: quality bytes on their way to d_qual + kQualPad + pf_off before their tile is pushed
    const uint8_t *pf_src = nullptr;
    uint64_t pf_n = 0, pf_off = 0;
    bool pf_active = false;
    cl_options opt{};
    Opts dopt{};
    std::string err;

    // host staging of the current contig
    bool in_contig = false, uploaded = false, ran = false;
    bool deep = false;               // this contig needs the 32-bit counter variant of k_pileup
    bool bits = true;                // the pass-bit form (default); DUT_QUAL_FORM=bytes at cl_create: the byte forms
    bool host_only = false;          // cl_debug_host_create: staging and the row builder only, for the CPU test suite
    int form = 0;                    // the form of k_pileup the resident contig gets (pick_form, at upload)
    uint32_t tune_ablate = 0;        // CL_TUNING builds: CL_ABLATE, read once at cl_create
    bool has_long = false;           // some read has more than kLongOps CIGAR ops (its checkpoints are in h_ck_x / h_ck_y)
    int32_t tid = 0;
    uint32_t contig_len = 0;
    RawVec<uint8_t> h_ref;
    RawVec<int32_t> h_pos;
    RawVec<uint8_t> h_mapq;
    RawVec<uint32_t> h_cigar_off;
    RawVec<uint32_t> h_cigar;
    RawVec<unsigned long long> h_qual_off;
    std::vector<uint8_t> h_qual;     // quality bytes of small tiles, not yet on the device
    uint64_t q_dev = 0;              // quality bytes of this contig that already are (d_qual + kQualPad ..)
    // reads whose reference span exceeds kWideSpan (ascending read index = ascending position)
    std::vector<uint32_t> h_wide_idx;
    // short-read form: read i's records are rec[h_rec_of[i] .. h_rec_of[i + 1]) (built at upload, gen_read_recs)
    RawVec<uint32_t> h_rec_of;            // (pooled with the other staging arrays: 4 bytes per read, written at every upload)
    std::vector<uint32_t> h_wide_rec_of;   // prefix sums of the wide reads' record counts (n_wide + 1 entries)
    uint32_t n_rec = 0;
    std::vector<int32_t> h_wide_pos;
    // the index over the CIGARs, built by the one host walk that validates a tile (cl_push_reads): every read's end,
    // and for reads with more than kLongOps operations the (reference, query) position before every 64th operation
    // of the contig's CIGAR array
    RawVec<uint32_t> h_end, h_ck_x, h_ck_y;
    // per read, from the same walk: how many records the record forms get for it (gen_read_recs); their prefix sums
    // are taken at upload
    RawVec<uint32_t> h_rec_cnt;
    // pass-bit form (the default): bit g = quality byte g of the contig passes min_base_quality (mod.rs:33), taken in
    // cl_push_reads' walk; and, from the same walk, the sum of the passing qualities over the M/=/X bases of the reads
    // with mapq >= min_mapping_quality (contig_profiler.rs:65-70: summed_baseq is per-read separable)
    RawVec<uint64_t> h_qbits;        // the reads' bit strings (reference order; pass_rows.h), word-aligned per read
    RawVec<unsigned long long> h_rb_off;   // n + 1 word offsets into h_qbits (| dut::kRowSparse)
    RawVec<uint32_t> h_sc_off, h_sc; // the CIGARs of the sparse reads (n + 1 offsets)
    uint64_t host_sum_q = 0;         // of the contig being pushed
    // ... and the reads' other separable sums (contig_profiler.rs:74, 79-82): reference spans of the reads
    // the pileup holds (-> summed_coverage) and mapq x span over those with mapq >= min_mapping_quality (-> summed_mapq)
    uint64_t host_sum_cov = 0, host_sum_mapq = 0;
    uint64_t dev_sum_cov = 0, dev_sum_mapq = 0;   // of the resident contig
    uint32_t head_span = kHeadSpanMax;   // most positions one head of k_pileup_rows spans (DUT_HEAD_SPAN: a test hook)
    uint64_t host_n_ops = 0;         // CIGAR operations pushed for it (pass-bit form: none is staged; for cl_contig_layout)
    uint64_t dev_sum_q = 0;          // of the resident contig (handed to the summary workgroup of every run)
    bool rec_counted = true;           // false: a tile of long-read shape skipped the count (cl_contig_upload makes up for it if the contig gets the short-read form after all)
    uint32_t n_long = 0;                 // reads with more than kLongOps operations
    uint32_t host_err = 0;               // kErrCigar / kErrRange found by that walk (reported by cl_contig_collect)
    uint32_t bounds_err = 0;             // kErrRange raised by the window bounds (reported by cl_contig_collect)
    uint32_t span_n = 0, span_w = 0; // longest span among the ordinary / the wide reads
    uint32_t n_wide = 0;

    // device residents
    DevBuf<int32_t> d_pos;
    DevBuf<uint8_t> d_mapq;
    DevBuf<uint8_t> d_qual;
    DevBuf<uint8_t> d_ref;
    DevBuf<uint32_t> d_end;
    DevBuf<ReadRec> d_rec;           // the records of the short-read form of k_pileup
    DevBuf<uint2> d_heads;           // pass-bit form: the heads k_pileup_rows reads, {pos, span | low << 31}
    DevBuf<uint32_t> d_refn;         // pass-bit form: bit p = the reference base at p is 'N' / 'n' or lies beyond the reference
    DevBuf<uint4> d_rows;            // pass-bit form: the windows' rows, groups of 4 rows x 64 blocks (1 KB each)
    uint64_t n_row_groups = 0;
    uint32_t max_groups = 0;         // most groups of any window: picks the number of counter planes of k_pileup_rows
    DevBuf<uint32_t> d_win_off, d_wide_idx;
    DevBuf<WinMeta> d_win;
    DevBuf<uint8_t> d_state;         // per-position states: allocated and written for debug dumps only
    DevBuf<uint16_t> d_runs;         // per window kT entries: run starts inside the window
    DevBuf<uint8_t> d_first_state, d_last_state;
    DevBuf<uint8_t> d_win_wide;      // per window: sticky "needs 16-bit counter fields" mark (k_pileup)
    DevBuf<WinPartial> d_winpart;
    DevBuf<FinPartial> d_fin;
    DevBuf<uint32_t> d_errflag;        // [0] error bits raised by the kernels of a run, [1] unused
    DevBuf<uint2> d_runtab;            // run-table form: the windows' match pieces (host_build_runs)
    uint64_t n_runtab = 0;
    DevBuf<uint32_t> d_lut;
    DevBuf<DevSummary> d_summary;
    DevBuf<Interval> d_iv;
    DevBuf<uint32_t> d_dbg;          // 3 * n_win * T

    uint32_t n_reads = 0;
    uint64_t n_cigar = 0, n_qual = 0;
    uint32_t extent = 0, n_win = 0;

    std::vector<cl_interval> h_iv;
    DevSummary h_sum{};

    // profiling
    bool profiling = false;
    static constexpr int kEvSets = 64;       // runs that can be in flight before events are read back
    hipEvent_t ev[kEvSets][CL_K_COUNT + 1] = {};
    bool ev_made = false;
    int ev_pending = 0;
    double ms[CL_K_COUNT] = {};
    uint64_t n_runs = 0;
    // the last cl_site_pileup: duration of its kernel (HIP events on the stream) and its algorithmic bytes
    hipEvent_t site_ev[2] = {nullptr, nullptr};
    double site_ms = 0.0;
    uint64_t site_bytes = 0;
    SiteResident site;
};

static void join_prealloc(cl_ctx *c) { if (c->prealloc.joinable()) c->prealloc.join(); }   // (cl_contig_reserve's helper thread)

static void swap_staging(cl_ctx *c, StagingSet &o)
{
    c->h_ref.swap(o.ref); c->h_mapq.swap(o.mapq); c->h_pos.swap(o.pos); c->h_cigar_off.swap(o.cigar_off);
    c->h_cigar.swap(o.cigar); c->h_end.swap(o.end); c->h_ck_x.swap(o.ck_x); c->h_ck_y.swap(o.ck_y); c->h_qual_off.swap(o.qual_off);
    c->h_qbits.swap(o.qbits); c->h_rec_cnt.swap(o.rec_cnt); c->h_rb_off.swap(o.rb_off); c->h_rec_of.swap(o.rec_of);
}
// a context without staging memory of its own takes a pooled set (cl_contig_begin) ...
static void take_staging(cl_ctx *c)
{
    if (c->h_pos.cap || c->h_cigar.cap || c->h_qual_off.cap || c->h_qbits.cap) return;
    std::unique_ptr<StagingSet> s;
    {
        std::lock_guard<std::mutex> g(g_staging_mu);
        if (g_staging.empty()) return;
        s = std::move(g_staging.back()); g_staging.pop_back();
    }
    swap_staging(c, *s);                                   // what the context had (nothing) is freed with s
}
// ... and gives its set back once the contig is on the device (cl_contig_upload); a full pool keeps the larger sets
static void give_staging(cl_ctx *c)
{
    std::unique_ptr<StagingSet> s(new (std::nothrow) StagingSet());
    if (!s) return;
    swap_staging(c, *s);
    s->ref.clear(); s->mapq.clear(); s->pos.clear(); s->cigar_off.clear(); s->cigar.clear(); s->end.clear();
    s->ck_x.clear(); s->ck_y.clear(); s->qual_off.clear(); s->qbits.clear(); s->rec_cnt.clear(); s->rb_off.clear(); s->rec_of.clear();
    std::lock_guard<std::mutex> g(g_staging_mu);
    if (g_staging.size() < kStagingSets) { g_staging.push_back(std::move(s)); return; }
    size_t small = 0;
    for (size_t i = 1; i < g_staging.size(); ++i) if (g_staging[i]->cigar_off.cap < g_staging[small]->cigar_off.cap) small = i;
    if (g_staging[small]->cigar_off.cap < s->cigar_off.cap) g_staging[small].swap(s);      // s (the smaller one) is freed
}

namespace {

cl_status fail(cl_ctx *c, cl_status s, const std::string &m)
{
    if (c) c->err = m;
    return s;
}

#define HIP_TRY(ctx, call)                                                                   \
    do {                                                                                     \
        hipError_t e__ = (call);                                                             \
        if (e__ != hipSuccess)                                                               \
            return fail(ctx, e__ == hipErrorOutOfMemory ? CL_ERR_NOMEM : CL_ERR_DEVICE,      \
                        std::string(#call) + ": " + hipGetErrorString(e__));                 \
    } while (0)

// constants of the byte-parallel threshold test (kernels.hip.h swar_ge7)
void make_ge_consts(uint8_t T, uint32_t &ge_add, uint32_t &ge_or, uint32_t &ge_and)
{
    uint32_t add;
    if (T == 0) { add = 0x80u; ge_or = 0xFFFFFFFFu; ge_and = 0xFFFFFFFFu; }
    else if (T <= 128) { add = 128u - T; ge_or = 0xFFFFFFFFu; ge_and = 0xFFFFFFFFu; }
    else { add = 256u - T; ge_or = 0u; ge_and = 0u; }
    ge_add = add * 0x01010101u;
}

// lut[raw] = smallest low_mapq_count with (low as f64 / raw as f64) > max_low_mapq_fraction
// (callable_profiler.rs:100-101), or 0xFFFFFFFF when no count <= raw qualifies.  Built with the
// same IEEE f64 divide and compare as the reference, so the device test `low >= lut[raw]` is
// exact by construction (the quotient is monotone in `low`).
void build_lut(double frac, std::vector<uint32_t> &lut)
{
    lut.assign(kLutSize, 0xFFFFFFFFu);
    for (uint32_t raw = 1; raw < kLutSize; ++raw) {
        double g = std::floor(frac * (double)raw) - 2.0;
        if (!(g == g)) continue;                 // NaN fraction: the comparison is never true
        if (g > (double)raw) continue;
        uint32_t low = g < 0.0 ? 0u : (uint32_t)g;
        while (low <= raw && !(((double)low / (double)raw) > frac)) ++low;
        if (low <= raw) lut[raw] = low;
    }
}

cl_status ensure_pins(cl_ctx *c)
{
    if (c->ring) return CL_OK;
    c->ring = acquire_ring(c->device);
    if (!c->ring) return fail(c, CL_ERR_DEVICE, "cannot create the pinned staging ring (hipHostMalloc)");
    return CL_OK;
}

// n bytes to `dst` through the ring: fill(off, len, out) writes the bytes [off, off + len) of the transfer into the
// pinned buffer `out`.  Chunks are dealt round-robin to the copier threads.  Returns at once; ring_finish joins.
template <class Fill>
cl_status ring_start(cl_ctx *c, uint8_t *dst, uint64_t n, Fill fill, uint64_t chunk_bytes = PinRing::kPinBytes, int want_threads = 0)
{
    cl_status s = ensure_pins(c);
    if (s != CL_OK) return s;
    PinRing *R = c->ring.get();
    c->ring_t0 = StageTimer::now();
    R->acquire(c);                                        // another context of this device may be using the ring
    c->ring_held = true;
    const uint64_t CH = chunk_bytes, nch = (n + CH - 1) / CH;
    // (plain copies saturate the link with DUT_COPY_THREADS buffers in flight; a fill that gathers small pieces is bound by
    // the fill and asks for all of the ring's pairs)
    const int T = want_threads > 0 ? std::min(want_threads, std::max(1, R->slots)) : PinRing::threads();
    const int nt = (int)std::min<uint64_t>((uint64_t)T, nch);
    for (int t = 0; t < PinRing::kCopyThreads; ++t) c->copy_err[t] = hipSuccess;
    c->crew_busy = true;
    R->crew.start(nt, [c, R, dst, n, fill, nch, CH, T](int t) {
            const bool timing = StageTimer().on;
            double t_fill = 0, t_issue = 0, t_wait = 0, t0 = timing ? StageTimer::now() : 0.0, ta;
            hipError_t e = hipSetDevice(c->device);
            int k = 0;
            for (uint64_t ch = (uint64_t)t; ch < nch && e == hipSuccess; ch += (uint64_t)T, ++k) {
                const int b = k & 1;
                const uint64_t off = ch * CH, len = std::min<uint64_t>(CH, n - off);
                if (timing) ta = StageTimer::now();
                if (k >= 2) e = hipEventSynchronize(R->pin_ev[t][b]);           // the buffer's previous transfer is done
                if (e != hipSuccess) break;
                if (timing) { const double tb = StageTimer::now(); t_wait += tb - ta; ta = tb; }
                fill(off, len, R->pin[t][b]);
                if (timing) { const double tb = StageTimer::now(); t_fill += tb - ta; ta = tb; }
                e = hipMemcpyAsync(dst + off, R->pin[t][b], len, hipMemcpyHostToDevice, R->copy_stream[t]);
                if (e == hipSuccess) e = hipEventRecord(R->pin_ev[t][b], R->copy_stream[t]);
                if (timing) t_issue += StageTimer::now() - ta;
            }
            // the thread's last transfers (one per buffer it used), waited for on their events
            if (timing) ta = StageTimer::now();
            hipError_t e2 = hipSuccess;
            for (int b = 0; b < 2 && b < k; ++b) { const hipError_t w = hipEventSynchronize(R->pin_ev[t][b]); if (e2 == hipSuccess) e2 = w; }
            c->copy_err[t] = e != hipSuccess ? e : e2;
            if (timing) {
                const double t1 = StageTimer::now();
                fprintf(stderr, "[dut-timing]       ring thread %d: %d buffers, started %.1f ms after the call, fill %.1f, issue %.1f, wait %.1f + %.1f ms\n", t, k,
                        (t0 - c->ring_t0) * 1e3, t_fill * 1e3, t_issue * 1e3, t_wait * 1e3, (t1 - ta) * 1e3);
            }
    });
    return CL_OK;
}

cl_status ring_finish(cl_ctx *c)
{
    if (c->crew_busy) { c->ring->crew.wait(); c->crew_busy = false; }   // the ring's threads are done with this transfer
    if (c->ring_held) { c->ring_held = false; c->ring->release(c); }
    for (int t = 0; t < PinRing::kCopyThreads; ++t) HIP_TRY(c, c->copy_err[t]);
    return CL_OK;
}

// plain bytes through the ring, start to finish
cl_status ring_copy(cl_ctx *c, void *dst, const void *src, uint64_t n)
{
    if (n == 0) return CL_OK;
    const uint8_t *s8 = static_cast<const uint8_t *>(src);
    cl_status s = ring_start(c, static_cast<uint8_t *>(dst), n, [s8](uint64_t off, uint64_t len, uint8_t *out) { memcpy(out, s8 + off, len); });
    if (s != CL_OK) return s;
    return ring_finish(c);
}

// a prefetch that was started and never claimed by a tile: wait for it, its bytes are simply overwritten later
void drop_prefetch(cl_ctx *c)
{
    if (!c->pf_active) return;
    (void)ring_finish(c);
    c->pf_active = false; c->pf_src = nullptr; c->pf_n = 0;
}

cl_status ensure_events(cl_ctx *c)
{
    if (c->ev_made) return CL_OK;
    for (int s = 0; s < cl_ctx::kEvSets; ++s)
        for (int i = 0; i <= CL_K_COUNT; ++i) HIP_TRY(c, hipEventCreate(&c->ev[s][i]));
    c->ev_made = true;
    return CL_OK;
}

cl_status harvest_events(cl_ctx *c)
{
    for (int s = 0; s < c->ev_pending; ++s) {
        HIP_TRY(c, hipEventSynchronize(c->ev[s][CL_K_COUNT]));
        // events 0,3,4: the per-read and per-window indexes are built on the host at upload (CL_K_PREP and CL_K_BOUNDS
        // stay 0; ev[1], ev[2] are unused)
        static const int from[CL_K_COUNT] = {-1, -1, 0, 3}, to[CL_K_COUNT] = {-1, -1, 3, 4};
        for (int i = 0; i < CL_K_COUNT; ++i) {
            if (from[i] < 0) continue;
            float t = 0.f;
            HIP_TRY(c, hipEventElapsedTime(&t, c->ev[s][from[i]], c->ev[s][to[i]]));
            c->ms[i] += t;
        }
        c->n_runs += 1;
    }
    c->ev_pending = 0;
    return CL_OK;
}

// which form of k_pileup a resident contig gets: by its shape, decided once at upload and kept in the context
// (kernels.hip.h: LONG = 0 records (short reads), 2 the run table (long reads))
int pick_form(const cl_ctx *c)
{
    if (c->bits) return 3;           // the pass-bit form: head records + rows, whatever the reads' shape
    int form = 0;
    // long-read shape (8 or more CIGAR operations per read on average): the host's walk leaves a table of match pieces
    // per window (the run-table form).  It also wins where the operation-parallel form of rounds 1-2 was used -- long
    // match runs, HiFi-like: 0.157 against 0.292 ms at 20 Mb with 800-base runs, 0.185 against 0.382 with 150-base runs
    // (profiles/r03_hifi_forms.txt) --, so every long-read shape gets it.
    if (c->n_reads && c->n_cigar >= 8ull * c->n_reads) form = 2;
#ifdef CL_TUNING
    if (const char *fl = getenv("CL_FORCE_LONG")) { const int f = atoi(fl); if (f == 0 || f == 2) form = f; }
#endif
    return form;
}

Reads device_reads(const cl_ctx *c)
{
    Reads R;
    R.pos = c->d_pos.p; R.mapq = c->d_mapq.p; R.qual = c->d_qual.p + kQualPad; R.n = c->n_reads;
    return R;
}

// The records of one read for the short-read form of k_pileup (kernels.hip.h: ReadRec), in order: put(k, rec) for
// k = 0 .. count - 1; returns the count.  What the reference's column walk sees of the read (mod.rs:22-37): it is in
// every column of [pos, end) -- the head record --, and the bases of its M/=/X operations that have a quality byte are
// tested against min_base_quality -- the head's own run and the piece records.  Reads below min_mapping_quality are
// only counted (mod.rs:25): head alone.  A read without a reference span is in no column: no record.
template <class Put>
inline uint32_t gen_read_recs(int32_t pos, uint32_t end, uint32_t mq, uint32_t min_mapq, const uint32_t *cig, uint32_t nops,
                              unsigned long long q0, unsigned long long ql, Put &&put, uint32_t *phase = nullptr)
{
    // *phase: (reference position - query offset) mod 16 of the first run that gets a record -- where the read's quality
    // bytes have to start, mod 16, for that run's 16-position units to be 16-byte aligned in memory
    if (phase) *phase = 0u;
    bool first_run = true;
    const uint32_t span = end - (uint32_t)pos;
    if (span == 0u) return 0u;
    ReadRec head;
    head.pos = pos; head.span = span; head.qual_lo = 0u; head.meta = mq | 0x100u;
    uint32_t k = 1;
    if (mq >= min_mapq) {
        unsigned long long xr = (uint32_t)pos, y = 0;
        for (uint32_t j = 0; j < nops && xr <= 0xFFFF0000ull; ++j) {
            const uint32_t cw = cig[j], op = cw & 15u, l = cw >> 4;
            if ((0x181u >> op) & 1u) {                                     // M = X
                const unsigned long long lq = y < ql ? std::min<unsigned long long>(ql - y, l) : 0ull;   // bases that have a quality byte
                for (unsigned long long off = 0; off < lq; off += 0xFFFFull) {
                    const uint32_t len = (uint32_t)std::min<unsigned long long>(lq - off, 0xFFFFull);
                    const unsigned long long px = xr + off;
                    if (px > 0xFFFF0000ull) break;                          // flagged kErrRange by cl_push_reads
                    if (first_run) { first_run = false; if (phase) *phase = (uint32_t)((px - (y + off)) & 15ull); }
                    if (k == 1u && !(head.meta >> 16) && px == (uint32_t)pos) {
                        head.qual_lo = (uint32_t)(q0 + y + off); head.meta |= len << 16;
                    } else {
                        ReadRec r;
                        r.pos = (int32_t)(uint32_t)px; r.span = 0u; r.qual_lo = (uint32_t)(q0 + y + off); r.meta = mq | (len << 16);
                        put(k++, r);
                    }
                }
                xr += l; y += l;
            } else if ((0x18Du >> op) & 1u) xr += l;                       // D N
            else if ((0x193u >> op) & 1u) y += l;                          // I S
        }
    }
    put(0u, head);
    return k;
}

// A read's share of summed_baseq (contig_profiler.rs:65-70): the sum of the quality bytes that pass min_base_quality
// over the bases of its M/=/X operations that have a quality byte (q[0, ql): the read's quality string).  Reads of few
// operations: run by run; reads of many (long reads, a run every ~15 bases): the whole string minus the inserted and
// clipped bases, so that the vector loop sees long stretches.
inline uint64_t read_pass_sum(const uint8_t *q, unsigned long long ql, const uint32_t *cig, uint32_t nops, uint8_t thr, int level)
{
    unsigned long long y = 0;
    uint64_t sum = 0;
    if (nops <= 8u) {
        for (uint32_t j = 0; j < nops; ++j) {
            const uint32_t cw = cig[j], op = cw & 15u, l = cw >> 4;
            if ((0x181u >> op) & 1u) { if (y < ql) sum += dut::qual_pass_sum(q + y, std::min<unsigned long long>(ql - y, l), thr, level); y += l; }
            else if ((0x193u >> op) & 1u) y += l;
        }
        return sum;
    }
    uint64_t minus = 0;
    for (uint32_t j = 0; j < nops; ++j) {
        const uint32_t cw = cig[j], op = cw & 15u, l = cw >> 4;
        if ((0x181u >> op) & 1u) y += l;
        else if ((0x193u >> op) & 1u) { if (y < ql) minus += dut::qual_pass_sum(q + y, std::min<unsigned long long>(ql - y, l), thr, 0); y += l; }
    }
    return dut::qual_pass_sum(q, std::min<unsigned long long>(ql, y), thr, level) - minus;
}

// DUT_FAULT_INJECT=<what> (tests only; read at every upload): the named builder hands the bounds checks of the upload an
// index that lies outside its array -- "rows" a window's group range, "runtab" a read's quality offset in the run table,
// "rec" a record's quality offset -- so that a test can see the check refuse what would otherwise be a device fault.
bool fault_injected(const char *what)
{
    const char *e = getenv("DUT_FAULT_INJECT");
    return e && strcmp(e, what) == 0;
}

// bytes of read records per pinned buffer (DUT_REC_CHUNK: a test hook that puts the buffer seams inside the records of
// one read with small inputs; a multiple of the record size; read once)
uint64_t rec_chunk_bytes()
{
    static const uint64_t n = [] {
        const char *e = getenv("DUT_REC_CHUNK");
        uint64_t v = e ? strtoull(e, nullptr, 0) : PinRing::kPinBytes;
        v &= ~(uint64_t)(sizeof(ReadRec) - 1);
        return v < sizeof(ReadRec) ? sizeof(ReadRec) : (v > PinRing::kPinBytes ? PinRing::kPinBytes : v);
    }();
    return n;
}

// h_rec_of: the prefix sums of the reads' record counts (counted by cl_push_reads' walk)
cl_status build_rec_index(cl_ctx *c)
{
    const size_t n = c->h_pos.size();
    RawVec<uint32_t> &ro = c->h_rec_of;
    ro.resize(n + 1);
    ro[0] = 0u;
    if (!c->rec_counted) {
        // some tile looked like long reads and skipped the count, yet the contig as a whole gets the short-read form
        const int32_t *hp = c->h_pos.data(); const uint8_t *hm = c->h_mapq.data(); const uint32_t *he = c->h_end.data();
        const uint32_t *hc = c->h_cigar_off.data(), *hcig = c->h_cigar.data(); const unsigned long long *hq = c->h_qual_off.data();
        const uint32_t min_mapq = c->opt.min_mapping_quality;
        uint32_t *cw = c->h_rec_cnt.data();
        dut::parallel_for(n, dut::grain_for(n, 65536), [&](size_t i) {
            cw[i] = gen_read_recs(hp[i], he[i], hm[i], min_mapq, hcig + hc[i], hc[i + 1] - hc[i], 0ull, hq[i + 1] - hq[i], [](uint32_t, const ReadRec &) {});
        });
        c->rec_counted = true;
    }
    const uint32_t *cnt = c->h_rec_cnt.data();
    const size_t grain = dut::grain_for(n, 262144), nchunk = n ? (n + grain - 1) / grain : 0;
    std::vector<uint64_t> tot(nchunk + 1, 0);
    dut::parallel_for(nchunk, 1, [&](size_t k) {
        const size_t a = k * grain, b = std::min(n, a + grain);
        uint64_t t = 0;
        for (size_t i = a; i < b; ++i) t += cnt[i];
        tot[k + 1] = t;
    });
    for (size_t k = 0; k < nchunk; ++k) tot[k + 1] += tot[k];
    if (tot[nchunk] >= (1ull << 29)) return fail(c, CL_ERR_RANGE, "more than 2^29 read records in one contig");
    dut::parallel_for(nchunk, 1, [&](size_t k) {
        const size_t a = k * grain, b = std::min(n, a + grain);
        uint32_t run = (uint32_t)tot[k];
        for (size_t i = a; i < b; ++i) { run += cnt[i]; ro[i + 1] = run; }
    });
    c->n_rec = (uint32_t)tot[nchunk];
    return CL_OK;
}

// The windows' candidate ranges: an index of the resident reads (binary searches over the sorted positions), built on
// the host at upload -- where the positions still are -- instead of in every run (round 1 ran the same rules as a
// device function in front of every pileup launch; the parity tests hold the results of this one against the oracle).
void host_window_bounds(const cl_ctx *c, std::vector<WinMeta> &win, uint32_t &flags)
{
    const uint32_t n = (uint32_t)c->h_pos.size(), n_wide = (uint32_t)c->h_wide_pos.size();
    const int32_t *pos = c->h_pos.data(), *wpos = c->h_wide_pos.data();
    auto lb = [](const int32_t *p, uint32_t cnt, long long key) {
        return (uint32_t)(std::lower_bound(p, p + cnt, key, [](int32_t v, long long k) { return (long long)v < k; }) - p);
    };
    std::atomic<uint32_t> fl{0};
    win.resize((size_t)c->n_win + 1);
    dut::parallel_for(c->n_win, 512, [&](size_t w) {
        const long long W = (long long)w * kT;
        WinMeta m;
        m.lo = lb(pos, n, W - (long long)c->span_n + 1);
        m.hi = lb(pos, n, W + (long long)kT);
        m.wlo = 0; m.wn = 0;
        if (n_wide) {
            m.wlo = lb(wpos, n_wide, W - (long long)c->span_w + 1);
            m.wn = lb(wpos, n_wide, W - (long long)c->span_n + 1) - m.wlo;
        }
        m.q0 = 0; m.rlo = 0; m.rn = 0;
        if (!c->bits) {
            // the byte forms of k_pileup address the quality bytes of a window with 32-bit offsets
            const uint32_t first = m.wn ? c->h_wide_idx[m.wlo] : m.lo;      // lo <= n: the offsets array has n + 1 entries
            const unsigned long long qf = c->h_qual_off[first], qh = c->h_qual_off[m.hi];
            m.q0 = qf;
            if (m.hi > first && qh - qf > 0xFFFF0000ull) fl.fetch_or(kErrRange);
        }
        win[w] = m;
    });
    flags = fl.load();
}

// ... and, once the host's walks over the reads of the windows are done (run table, rows: they index reads), the ranges as
// the kernel wants them: the record forms' candidates are records -- those of the reads [lo, hi) lie side by side, the
// wide reads' are listed in wide_rec (wro: the prefix sums of the wide reads' record counts).
void finish_windows(const cl_ctx *c, const std::vector<uint32_t> &wro_v, std::vector<WinMeta> &win, uint32_t &flags)
{
    static const uint32_t kZero[1] = {0u};
    const uint32_t *wro = wro_v.empty() ? kZero : wro_v.data();      // (no wide read: wlo = wn = 0 everywhere)
    std::atomic<uint32_t> fl{0};
    dut::parallel_for(c->n_win, 4096, [&](size_t w) {
        WinMeta &m = win[w];
        if (c->form != 2) {
            const uint32_t *ro = c->h_rec_of.data();
            m.lo = ro[m.lo]; m.hi = ro[m.hi];
            const uint32_t w1 = wro[m.wlo + m.wn];
            m.wlo = wro[m.wlo]; m.wn = w1 - m.wlo;
        }
        // more candidates than the 16-bit differences of the pileup kernels can hold: the 32-bit variant is needed
        if ((m.hi - m.lo) + m.wn > 32767u) fl.fetch_or(kNeedDeep);
    });
    flags |= fl.load();
}


// The run table of the run-table form (kernels.hip.h, LONG = 2): per window of kT positions the M/=/X pieces of the reads
// that cover it -- what the reference's column walk visits as (alignment, qpos) with !is_del (mod.rs:30-37), grouped by
// window instead of by column.  One more walk over the staged CIGARs, at upload: a thread takes a range of windows
// and sweeps it with a list of read cursors (operation index, reference and query position), so every operation is
// visited once per range it touches; reads that start before the range enter at their last checkpoint in front of it
// (the 64-operation checkpoints of cl_push_reads' walk).  Reads below min_mapping_quality never enter (mod.rs:25).
// A piece = {x, y}: x + 16 u = the byte offset of unit u's qualities from the window's quality base;
// y = start | end - 1 << 11 | (read & 1) << 29 | valid << 31; it covers the unit of its start and at most the next one.
// The pieces are written straight into the pinned buffers of the staging ring and leave for HBM as a buffer fills: the
// table exists nowhere in host memory.  Buffers are placed in the device array in the order they fill (a window only
// needs its own pieces contiguous: its record holds their absolute index).
struct RunCur { uint32_t k, k1, x, y, qlen, flags; unsigned long long q0; };

// One window: the cursors of `act` emit their pieces inside [W, W + kT) to out[0, cap) and move on; finished reads
// leave the list.  Returns the number of pieces, or SIZE_MAX when `cap` did not suffice (the list is then spoilt: the
// caller restores its copy).
// `qend` = bytes of the padded quality allocation: every piece is checked against it as it is emitted -- the kernel loads
// 16 bytes at allocation offset qwin + (uint32)(x + 16 u) for the unit(s) of the piece, and the lanes behind a window's
// last entry repeat that entry's loads -- and *out_of_range is set when a load would leave the allocation (the upload then
// fails with CL_ERR_RANGE instead of launching a kernel that faults).
size_t sweep_window(std::vector<RunCur> &act, const uint32_t *cig, uint32_t W, unsigned long long qwin, uint2 *out, size_t cap,
                    unsigned long long qend, std::atomic<bool> *out_of_range)
{
    const uint32_t Wend = W + kT;
    size_t n = 0, keep = 0;
    const size_t na = act.size();
    for (size_t i = 0; i < na; ++i) {
        RunCur cu = act[i];
        // every piece of an operation that starts at (x, y) has the same first word: q + kQualPad - sr with q = qrel + y +
        // (sp - x) and sr = sp - W
        const uint32_t qb = (uint32_t)(cu.q0 - qwin) + (uint32_t)kQualPad + W;
        while (cu.k < cu.k1 && cu.x < Wend) {
            const uint32_t cw = cig[cu.k], op = cw & 15u, l = cw >> 4;
            const uint32_t radv = (0x18Du >> op) & 1u, qadv = (0x193u >> op) & 1u, ism = (0x181u >> op) & 1u;
            const uint32_t xe = cu.x + (radv ? l : 0u);
            if (ism) {
                if (n + (kT / 32u + 2u) > cap) return SIZE_MAX;                 // what one clipped run can emit at most
                const uint32_t sp = cu.x > W ? cu.x : W;
                const uint32_t lq = cu.y < cu.qlen ? std::min(cu.qlen - cu.y, l) : 0u;   // bases that have a quality byte
                uint32_t tp = xe < Wend ? xe : Wend;
                tp = (cu.x + lq) < tp ? (cu.x + lq) : tp;
                if (sp < tp) {
                    uint32_t sr = sp - W;
                    const uint32_t tr = tp - W, ex = qb + cu.y - cu.x;
                    do {
                        const uint32_t pe = std::min(tr, ((sr >> 4) + 2u) << 4);
                        out[n++] = make_uint2(ex, sr | ((pe - 1u) << 11) | cu.flags);
                        // the first and the last unit the kernel loads for this piece, as it computes their addresses
                        if (qwin + (uint32_t)(ex + ((sr >> 4) << 4)) + 16ull > qend || qwin + (uint32_t)(ex + (((pe - 1u) >> 4) << 4)) + 16ull > qend)
                            out_of_range->store(true, std::memory_order_relaxed);
                        sr = pe;
                    } while (sr < tr);
                }
            }
            if (xe > Wend) break;                                // the operation goes on in the next window
            if (xe < cu.x) { cu.k = cu.k1; break; }              // wraps the 32-bit coordinate: flagged kErrRange by cl_push_reads
            cu.x = xe; cu.y += qadv ? l : 0u; cu.k += 1u;
        }
        if (cu.k < cu.k1) act[keep++] = cu;
    }
    act.resize(keep);
    return n;
}

// pieces per pinned buffer (DUT_RUN_CHUNK: a test hook that makes the buffer-full and oversized-window paths reachable
// with small inputs; read once)
size_t run_chunk_entries()
{
    static const size_t n = [] {
        const char *e = getenv("DUT_RUN_CHUNK");
        const size_t full = PinRing::kPinBytes / sizeof(uint2);
        const size_t v = e ? (size_t)strtoull(e, nullptr, 0) : full;
        return v < kT / 32u + 2u ? kT / 32u + 2u : (v > full ? full : v);
    }();
    return n;
}

cl_status stream_run_table(cl_ctx *c, std::vector<WinMeta> &win)
{
    const uint32_t n_win = c->n_win;
    const int32_t *pos = c->h_pos.data();
    const uint8_t *mapq = c->h_mapq.data();
    const uint32_t *end = c->h_end.data(), *coff = c->h_cigar_off.data(), *cig = c->h_cigar.data();
    const uint32_t *ckx = c->h_ck_x.data(), *cky = c->h_ck_y.data(), *wide_idx = c->h_wide_idx.data();
    const unsigned long long *qoff = c->h_qual_off.data();
    const uint32_t min_mapq = c->opt.min_mapping_quality;
    c->n_runtab = 0;
    if (n_win == 0) return CL_OK;
    cl_status s = ensure_pins(c);
    if (s != CL_OK) return s;
    // (as many walkers as the staging ring has buffer pairs -- DUT_COPY_THREADS, 8 by default --: pinning 16 MB more per
    // further walker costs more than the walker saves: 73 ms with 16 walkers against 23 ms with 8 at chr21 30x)
    const int nt = std::max(1, std::min<int>(std::max(1, c->ring->slots), dut::worker_threads()));
    // tasks: several per thread so that uneven depth evens out, not so short that the range-start walks show
    const size_t per = std::max<size_t>(16, (size_t)n_win / (8 * (size_t)nt) + 1);
    const size_t ntasks = ((size_t)n_win + per - 1) / per;
    const size_t capE = run_chunk_entries();
    const unsigned long long qend = c->n_qual + 2ull * kQualPad;
    std::atomic<bool> oor{false};
    const bool inject = fault_injected("runtab");                // test hook: one read's quality offset is moved out of range
    // the device array: an estimate first (a piece per ~12 aligned bases); a contig that needs more tells how much
    uint64_t want = c->n_qual / 12 + (uint64_t)n_win * 8 + 65536;
    for (int attempt = 0; attempt < 2; ++attempt) {
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, c->d_runtab.reserve(want));
        uint2 *const d_tab = c->d_runtab.p;
        const uint64_t dev_cap = c->d_runtab.cap;
        std::atomic<uint64_t> dev_next{0};
        std::atomic<size_t> next_task{0};
        PinRing *R = c->ring.get();
        R->acquire(c);
        c->ring_held = true;
        if (!R->ensure_slots(nt)) { (void)ring_finish(c); return fail(c, CL_ERR_DEVICE, "cannot extend the pinned staging ring (hipHostMalloc)"); }
        for (int t = 0; t < PinRing::kCopyThreads; ++t) c->copy_err[t] = hipSuccess;
        c->crew_busy = true;
        R->crew.start(nt, [&](int t) {
                hipError_t err = hipSetDevice(c->device);
                int kb = 0;                                                  // buffers this thread has sent
                auto cur_buf = [&]() { return reinterpret_cast<uint2 *>(R->pin[t][kb & 1]); };
                auto send = [&](uint2 *dst, size_t cnt) {                    // the current buffer leaves; on to the other one
                    if (err != hipSuccess) return;
                    err = hipMemcpyAsync(dst, cur_buf(), cnt * sizeof(uint2), hipMemcpyHostToDevice, R->copy_stream[t]);
                    if (err == hipSuccess) err = hipEventRecord(R->pin_ev[t][kb & 1], R->copy_stream[t]);
                    ++kb;
                    if (kb >= 2 && err == hipSuccess) err = hipEventSynchronize(R->pin_ev[t][kb & 1]);   // its previous transfer is done
                };
                std::vector<RunCur> act, save;
                std::vector<uint32_t> in_buf;                                // windows whose pieces lie in the current buffer
                RawVec<uint2> big;
                size_t used = 0;
                auto flush = [&]() {
                    if (!used) return;
                    const uint64_t off = dev_next.fetch_add(used);
                    if (off + used <= dev_cap) send(d_tab + off, used);      // else: the array is too small, only the total counts now
                    for (uint32_t w : in_buf) win[w].rlo += (uint32_t)off;
                    in_buf.clear(); used = 0;
                };
                auto enter = [&](uint32_t r, uint32_t W) {                   // a read that covers positions at or after W
                    if (mapq[r] < min_mapq) return;
                    RunCur cu;
                    cu.k = coff[r]; cu.k1 = coff[r + 1]; cu.x = (uint32_t)pos[r]; cu.y = 0;
                    if (cu.k >= cu.k1 || end[r] <= W) return;
                    const unsigned long long ql = qoff[r + 1] - qoff[r];
                    cu.qlen = ql > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)ql;
                    cu.q0 = qoff[r]; cu.flags = ((r & 1u) << 29) | 0x80000000u;
                    if (inject && r == c->n_reads / 2u) cu.q0 += 0x7FFFFFF0ull;
                    if (cu.k1 - cu.k > kLongOps && cu.x < W) {               // the last checkpoint at or before W
                        const uint32_t jlo = (cu.k + 63u) >> 6, jhi = (cu.k1 - 1u) >> 6;
                        if (jlo <= jhi && ckx[jlo] <= W) {
                            uint32_t lo_j = jlo, hi_j = jhi;
                            while (lo_j < hi_j) {
                                const uint32_t mid = lo_j + ((hi_j - lo_j + 1u) >> 1);
                                if (ckx[mid] <= W) lo_j = mid; else hi_j = mid - 1u;
                            }
                            cu.k = lo_j << 6; cu.x = ckx[lo_j]; cu.y = cky[lo_j];
                        }
                    }
                    act.push_back(cu);
                };
                try {
                    size_t task;
                    while ((task = next_task.fetch_add(1)) < ntasks) {
                        const size_t w0 = task * per, w1 = std::min<size_t>(n_win, w0 + per);
                        act.clear();
                        for (size_t w = w0; w < w1; ++w) {
                            const uint32_t W = (uint32_t)(w * kT);
                            WinMeta &m = win[w];
                            if (w == w0) {
                                // what covers the range's first window: the wide reads in front of read lo, then [lo, hi)
                                for (uint32_t i = 0; i < m.wn; ++i) enter(wide_idx[m.wlo + i], W);
                                for (uint32_t r = m.lo; r < m.hi; ++r) enter(r, W);
                            } else {
                                for (uint32_t r = win[w - 1].hi; r < m.hi; ++r) enter(r, W);   // the reads that start in this window
                            }
                            save = act;
                            size_t cnt = sweep_window(act, cig, W, m.q0, cur_buf() + used, capE - used, qend, &oor);
                            if (cnt == SIZE_MAX) {                           // the buffer is full: it leaves, the window starts over
                                flush();
                                act = save;
                                cnt = sweep_window(act, cig, W, m.q0, cur_buf(), capE, qend, &oor);
                            }
                            if (cnt == SIZE_MAX) {
                                // a window that no buffer holds (thousandfold depth): through a block of its own
                                size_t bc = capE * 4;
                                for (;;) {
                                    big.clear(); big.resize(bc);
                                    act = save;
                                    cnt = sweep_window(act, cig, W, m.q0, big.data(), bc, qend, &oor);
                                    if (cnt != SIZE_MAX) break;
                                    bc *= 4;
                                }
                                const uint64_t off = dev_next.fetch_add(cnt);
                                if (off + cnt <= dev_cap && err == hipSuccess)
                                    err = hipMemcpy(d_tab + off, big.data(), cnt * sizeof(uint2), hipMemcpyHostToDevice);
                                m.rlo = (uint32_t)off; m.rn = (uint32_t)std::min<size_t>(cnt, 0xFFFFFFFFu);
                                continue;
                            }
                            m.rlo = (uint32_t)used; m.rn = (uint32_t)cnt;
                            if (cnt) in_buf.push_back((uint32_t)w);
                            used += cnt;
                        }
                    }
                    flush();
                } catch (...) { if (err == hipSuccess) err = hipErrorOutOfMemory; }
                for (int b = 0; b < 2 && b < kb; ++b) { const hipError_t e = hipEventSynchronize(R->pin_ev[t][b]); if (err == hipSuccess) err = e; }
                c->copy_err[t] = err;
        });
        s = ring_finish(c);
        if (s != CL_OK) return s;
        const uint64_t total = dev_next.load();
        if (oor.load()) return fail(c, CL_ERR_RANGE, "a match piece of the run table addresses quality bytes outside the resident array");
        if (total >= 0xFFFFFFF0ull) return fail(c, CL_ERR_RANGE, "more than 2^32 match pieces in one contig");
        if (total <= dev_cap) { c->n_runtab = total; return CL_OK; }
        want = total;                                            // exact now: once more
    }
    return fail(c, CL_ERR_DEVICE, "run table: the second sizing pass did not fit");
}

// groups of rows per pinned buffer (DUT_ROW_CHUNK: a test hook that makes the buffer-full and oversized-window paths
// reachable with small inputs; read once)
size_t row_chunk_groups()
{
    static const size_t n = [] {
        const char *e = getenv("DUT_ROW_CHUNK");
        const size_t full = PinRing::kPinBytes / (dut::kRowGroupWords * sizeof(uint32_t));
        const size_t v = e ? (size_t)strtoull(e, nullptr, 0) : full;
        return v < 1 ? 1 : (v > full ? full : v);
    }();
    return n;
}

dut::RowReads row_reads(const cl_ctx *c)
{
    dut::RowReads H;
    H.pos = c->h_pos.data(); H.end = c->h_end.data(); H.mapq = c->h_mapq.data();
    H.off = c->h_rb_off.data(); H.bits = c->h_qbits.data();
    H.sc_off = c->h_sc_off.data(); H.sc = c->h_sc.data();
    H.min_mapq = c->opt.min_mapping_quality;
    return H;
}

// The pass-bit rows of the resident contig (pass_rows.h), laid out at upload from the reads' reference-order bit strings
// (no CIGAR is looked at but those of the few gapped reads) and streamed to HBM the way the run table is: a thread takes
// a range of windows and sweeps it with a list of read cursors;
// the groups of a window are written straight into the pinned buffers of the staging ring, a buffer leaves when the
// next window no longer fits, buffers are placed in the device array in the order they fill (a window only needs its
// own groups contiguous: its record holds their index).  win[w].rlo / rn = first group / number of groups.
cl_status stream_rows(cl_ctx *c, std::vector<WinMeta> &win)
{
    const uint32_t n_win = c->n_win;
    const dut::RowReads H = row_reads(c);
    const uint32_t *wide_idx = c->h_wide_idx.data();
    c->n_row_groups = 0; c->max_groups = 0;
    if (n_win == 0) return CL_OK;
    cl_status s = ensure_pins(c);
    if (s != CL_OK) return s;
    // (as many walkers as the staging ring has buffer pairs -- DUT_COPY_THREADS, 8 by default --: pinning 16 MB more per
    // further walker costs more than the walker saves: 73 ms with 16 walkers against 23 ms with 8 at chr21 30x)
    const int nt = std::max(1, std::min<int>(std::max(1, c->ring->slots), dut::worker_threads()));
    const size_t per = std::max<size_t>(16, (size_t)n_win / (8 * (size_t)nt) + 1);
    const size_t ntasks = ((size_t)n_win + per - 1) / per;
    const size_t capG = row_chunk_groups();
    constexpr size_t GW = dut::kRowGroupWords;
    // the device array: an estimate first (rows ~ 1.7 x the mean depth, a quarter of that in groups, one group of
    // rounding per window); a contig that needs more tells how much
    uint64_t want = (c->n_qual / kT) * 17 / 40 + (uint64_t)n_win + 1024;
    for (int attempt = 0; attempt < 2; ++attempt) {
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, c->d_rows.reserve(want * (GW / 4)));
        uint32_t *const d_tab = reinterpret_cast<uint32_t *>(c->d_rows.p);
        const uint64_t dev_cap = c->d_rows.cap / (GW / 4);              // groups
        std::atomic<uint64_t> dev_next{0};
        std::atomic<size_t> next_task{0};
        std::atomic<uint32_t> max_groups{0};
        PinRing *R = c->ring.get();
        R->acquire(c);
        c->ring_held = true;
        if (!R->ensure_slots(nt)) { (void)ring_finish(c); return fail(c, CL_ERR_DEVICE, "cannot extend the pinned staging ring (hipHostMalloc)"); }
        for (int t = 0; t < PinRing::kCopyThreads; ++t) c->copy_err[t] = hipSuccess;
        c->crew_busy = true;
        R->crew.start(nt, [&](int t) {
                hipError_t err = hipSetDevice(c->device);
                int kb = 0;                                                  // buffers this thread has sent
                auto cur_buf = [&]() { return reinterpret_cast<uint32_t *>(R->pin[t][kb & 1]); };
                auto send = [&](uint32_t *dst, size_t groups) {              // the current buffer leaves; on to the other one
                    if (err != hipSuccess) return;
                    err = hipMemcpyAsync(dst, cur_buf(), groups * GW * sizeof(uint32_t), hipMemcpyHostToDevice, R->copy_stream[t]);
                    if (err == hipSuccess) err = hipEventRecord(R->pin_ev[t][kb & 1], R->copy_stream[t]);
                    ++kb;
                    if (kb >= 2 && err == hipSuccess) err = hipEventSynchronize(R->pin_ev[t][kb & 1]);   // its previous transfer is done
                };
                std::vector<dut::RowCur> act, save;
                std::vector<uint32_t> in_buf;                                // windows whose groups lie in the current buffer
                dut::RowScratch sc;
                RawVec<uint32_t> big;
                size_t used = 0;
                uint32_t my_max = 0;
                auto flush = [&]() {
                    if (!used) return;
                    const uint64_t off = dev_next.fetch_add(used);
                    if (off + used <= dev_cap) send(d_tab + off * GW, used);  // else: the array is too small, only the total counts now
                    for (uint32_t w : in_buf) win[w].rlo += (uint32_t)off;
                    in_buf.clear(); used = 0;
                };
                try {
                    size_t task;
                    while ((task = next_task.fetch_add(1)) < ntasks) {
                        const size_t w0 = task * per, w1 = std::min<size_t>(n_win, w0 + per);
                        act.clear();
                        for (size_t w = w0; w < w1; ++w) {
                            const uint32_t W = (uint32_t)(w * kT);
                            WinMeta &m = win[w];
                            if (w == w0) {
                                // what covers the range's first window: the wide reads in front of read lo, then [lo, hi)
                                for (uint32_t i = 0; i < m.wn; ++i) dut::rows_enter(act, H, wide_idx[m.wlo + i], W);
                                for (uint32_t r = m.lo; r < m.hi; ++r) dut::rows_enter(act, H, r, W);
                            } else {
                                for (uint32_t r = win[w - 1].hi; r < m.hi; ++r) dut::rows_enter(act, H, r, W);   // the reads that start in this window
                            }
                            if (act.empty()) { m.rlo = 0; m.rn = 0; continue; }
                            save = act;
                            size_t cnt = dut::rows_window<kT>(act, H, W, cur_buf() + used * GW, capG - used, sc);
                            if (cnt == SIZE_MAX) {                           // the buffer is full: it leaves, the window starts over
                                flush();
                                act = save;
                                cnt = dut::rows_window<kT>(act, H, W, cur_buf(), capG, sc);
                            }
                            if (cnt == SIZE_MAX) {
                                // a window that no buffer holds (depth in the ten thousands): through a block of its own
                                size_t bc = capG * 4;
                                for (;;) {
                                    big.clear(); big.resize(bc * GW);
                                    act = save;
                                    cnt = dut::rows_window<kT>(act, H, W, big.data(), bc, sc);
                                    if (cnt != SIZE_MAX) break;
                                    bc *= 4;
                                }
                                const uint64_t off = dev_next.fetch_add(cnt);
                                if (off + cnt <= dev_cap && err == hipSuccess)
                                    err = hipMemcpy(d_tab + off * GW, big.data(), cnt * GW * sizeof(uint32_t), hipMemcpyHostToDevice);
                                m.rlo = (uint32_t)off; m.rn = (uint32_t)std::min<size_t>(cnt, 0xFFFFFFFFu);
                                my_max = std::max(my_max, m.rn);
                                continue;
                            }
                            m.rlo = (uint32_t)used; m.rn = (uint32_t)cnt;
                            my_max = std::max(my_max, m.rn);
                            if (cnt) in_buf.push_back((uint32_t)w);
                            used += cnt;
                        }
                    }
                    flush();
                } catch (...) { if (err == hipSuccess) err = hipErrorOutOfMemory; }
                for (int b = 0; b < 2 && b < kb; ++b) { const hipError_t e = hipEventSynchronize(R->pin_ev[t][b]); if (err == hipSuccess) err = e; }
                uint32_t seen = max_groups.load();
                while (seen < my_max && !max_groups.compare_exchange_weak(seen, my_max)) {}
                c->copy_err[t] = err;
        });
        s = ring_finish(c);
        if (s != CL_OK) return s;
        const uint64_t total = dev_next.load();
        if (total >= 0xFFFFFFF0ull) return fail(c, CL_ERR_RANGE, "more than 2^32 groups of pass-bit rows in one contig");
        if (total <= dev_cap) { c->n_row_groups = total; c->max_groups = max_groups.load(); return CL_OK; }
        want = total;                                            // exact now: once more
    }
    return fail(c, CL_ERR_DEVICE, "pass-bit rows: the second sizing pass did not fit");
}

// allocate and lay out everything that depends on the extent (called by cl_contig_upload, the staged arrays still there)
cl_status size_for_extent(cl_ctx *c, uint32_t extent)
{
    c->extent = extent;
    c->n_win = (uint32_t)(((uint64_t)extent + kT - 1) / kT);
    const size_t padded = (size_t)c->n_win * kT;
    HIP_TRY(c, c->d_win.reserve(c->n_win + 1));
    HIP_TRY(c, c->d_win_off.reserve(c->n_win + 1));
    HIP_TRY(c, c->d_winpart.reserve(c->n_win + 1));
    HIP_TRY(c, c->d_fin.reserve(c->n_win / kFinBlock + 2));
    HIP_TRY(c, c->d_runs.reserve(padded + 16));
    HIP_TRY(c, c->d_first_state.reserve(c->n_win + 1));
    HIP_TRY(c, c->d_last_state.reserve(c->n_win + 1));
    HIP_TRY(c, c->d_win_wide.reserve(c->n_win + 1));
    HIP_TRY(c, hipMemsetAsync(c->d_win_wide.p, 0, c->n_win + 1, c->stream));
    // reference bytes: [0,ref_len) from the caller, 'N' beyond (mod.rs:79-80).  The pass-bit form needs one bit of a
    // base -- is it 'N' / 'n' (mod.rs:100-101) --, taken here, where the bytes pass through the host's hands anyway:
    // 1/8 of the transfer, 1/8 of what every run reads
    if (c->form == 3) HIP_TRY(c, c->d_refn.reserve(padded / 32 + 4)); else HIP_TRY(c, c->d_ref.reserve(padded + 16));
    {
        const uint8_t *ref = c->h_ref.data();
        const uint64_t nref = std::min<uint64_t>(c->h_ref.size(), padded);
        cl_status rs;
        if (c->form == 3)
            rs = ring_start(c, reinterpret_cast<uint8_t *>(c->d_refn.p), padded / 8, [ref, nref](uint64_t off, uint64_t len, uint8_t *out) {
                // (a buffer of the ring is a whole number of 64-bit words: 4 MB, and padded / 8 = 256 bytes per window)
                const uint64_t p = off * 8;
                dut::ref_n_words(ref + std::min<uint64_t>(p, nref), p < nref ? nref - p : 0, len / 8, reinterpret_cast<uint64_t *>(out), dut::qual_pack_level());
            });
        else rs = ring_start(c, c->d_ref.p, padded + 16, [ref, nref](uint64_t off, uint64_t len, uint8_t *out) {
            const uint64_t have = off < nref ? std::min<uint64_t>(len, nref - off) : 0;
            if (have) memcpy(out, ref + off, have);
            if (have < len) memset(out + have, 'N', len - have);
        });
        // ... beside it, the windows' candidate ranges
        std::vector<WinMeta> win;
        uint32_t flags = 0;
        host_window_bounds(c, win, flags);
        if (rs == CL_OK) rs = ring_finish(c); else (void)ring_finish(c);
        if (rs != CL_OK) return rs;
        c->n_runtab = 0; c->n_row_groups = 0; c->max_groups = 0;
        if (c->form == 3) {
            // the windows' pass-bit rows: one more walk over the staged CIGARs, streamed to HBM through the ring
            StageTimer tr;
            rs = stream_rows(c, win);
            if (rs != CL_OK) return rs;
            tr.lap("upload: pass-bit rows (walk + H2D)");
        }
        if (c->form == 2 && !(flags & kErrRange)) {
            // the windows' match pieces: one more walk over the staged CIGARs, streamed to HBM through the ring
            StageTimer tr;
            rs = stream_run_table(c, win);
            if (rs != CL_OK) return rs;
            tr.lap("upload: run table (walk + H2D)");
        }
        // DUT_VALIDATE=1 (tooling: tools/fuzz_parity.py sets it): what the kernels will index is checked on the host before
        // anything is launched -- candidate ranges against the resident arrays, and every entry of the run table (read
        // back from the device) against the quality array -- so that a bad index is an error message, not a GPU fault
        finish_windows(c, c->h_wide_rec_of, win, flags);
        // what k_pileup_rows streams per window must lie inside the resident rows: checked in the product build, once per
        // contig (an index past the array is an error return, not a device fault)
        if (c->form == 3) {
            std::atomic<bool> bad{false};
            const uint64_t ngr = c->n_row_groups;
            if (fault_injected("rows") && c->n_win) win[c->n_win / 2].rlo += 0x7FFFFFF0u;
            dut::parallel_for(c->n_win, 8192, [&](size_t w) { if ((uint64_t)win[w].rlo + win[w].rn > ngr) bad.store(true); });
            if (bad.load()) return fail(c, CL_ERR_RANGE, "a window's pass-bit rows lie outside the resident row array");
        }
        static const bool validate = [] { const char *e = getenv("DUT_VALIDATE"); return e && *e == '1'; }();
        if (validate && !(flags & kErrRange)) {
            const uint64_t n_cand = c->form != 2 ? c->n_rec : c->h_pos.size();
            uint64_t n_wide_list = c->h_wide_idx.size();
            if (c->form != 2) n_wide_list = c->h_wide_rec_of.empty() ? 0 : c->h_wide_rec_of.back();
            std::vector<uint2> tab;
            if (c->form == 2 && c->n_runtab) {
                tab.resize(c->n_runtab);
                HIP_TRY(c, hipMemcpy(tab.data(), c->d_runtab.p, c->n_runtab * sizeof(uint2), hipMemcpyDeviceToHost));
            }
            const uint64_t qend = c->n_qual + 2 * (uint64_t)kQualPad;
            for (uint32_t w = 0; w < c->n_win; ++w) {
                const WinMeta &m = win[w];
                char msg[256];
                if (m.lo > m.hi || m.hi > n_cand || (uint64_t)m.wlo + m.wn > n_wide_list) {
                    snprintf(msg, sizeof(msg), "validate: window %u: candidates [%u, %u) of %llu, wide [%u, +%u) of %llu", w, m.lo, m.hi, (unsigned long long)n_cand, m.wlo, m.wn, (unsigned long long)n_wide_list);
                    return fail(c, CL_ERR_DEVICE, msg);
                }
                if (c->form != 2) continue;
                if ((uint64_t)m.rlo + m.rn > c->n_runtab) {
                    snprintf(msg, sizeof(msg), "validate: window %u: run table entries [%u, +%u) of %llu", w, m.rlo, m.rn, (unsigned long long)c->n_runtab);
                    return fail(c, CL_ERR_DEVICE, msg);
                }
                for (uint32_t i = 0; i < m.rn; ++i) {
                    const uint2 d = tab[(size_t)m.rlo + i];
                    const uint32_t sr = d.y & 2047u, er = (d.y >> 11) & 2047u, u0 = sr >> 4, u1 = er >> 4;
                    // the kernel adds the unit's 16 u to x in 32 bits and that to the window's base (allocation-relative: q0)
                    const unsigned long long off = m.q0 + (uint32_t)(d.x + (u1 << 4)) + 16ull;   // end of the last byte it loads for the entry
                    if (!(d.y >> 31) || er < sr || u1 > u0 + 1u || off > qend || m.q0 + (uint32_t)(d.x + (u0 << 4)) + 16ull > qend) {
                        snprintf(msg, sizeof(msg), "validate: window %u entry %u: x %u y 0x%08x (start %u end-1 %u), q0 %llu: loads up to byte %llu of %llu",
                                 w, i, d.x, d.y, sr, er, (unsigned long long)m.q0, off, (unsigned long long)qend);
                        return fail(c, CL_ERR_DEVICE, msg);
                    }
                }
            }
        }
        if (c->n_win) HIP_TRY(c, hipMemcpyAsync(c->d_win.p, win.data(), (size_t)c->n_win * sizeof(WinMeta), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        // a window with more candidates than the 16-bit counters / differences hold: the 32-bit form from the start
        c->deep = (flags & kNeedDeep) != 0;
        c->bounds_err = flags & kErrRange;
    }
    return CL_OK;
}

// what runs behind the pileup kernel: the windows' run lists -> intervals (scan + reduction over the windows, then one
// wave per window and an extra workgroup for the contig summary)
void launch_tail(cl_ctx *c)
{
    const uint32_t n_fin = (c->n_win + kFinBlock - 1) / kFinBlock;
    const uint32_t cap = (uint32_t)std::min<size_t>(c->d_iv.cap, 0xFFFFFFFFu);
    const unsigned long long sq = c->form == 3 ? c->dev_sum_q : 0ull, sc = c->form == 3 ? c->dev_sum_cov : 0ull, sm = c->form == 3 ? c->dev_sum_mapq : 0ull;
    if (n_fin)
        hipLaunchKernelGGL(k_fin_windows, dim3(n_fin), dim3(kFinBlock), 0, c->stream, c->d_winpart.p, c->d_first_state.p,
                           c->d_last_state.p, kT, c->n_win, c->extent, c->d_win_off.p, c->d_fin.p);
    hipLaunchKernelGGL((k_rle_write<(int)kT>), dim3((c->n_win + kBlock / 64 - 1) / (kBlock / 64) + 1), dim3(kBlock), 0, c->stream,
                       c->d_runs.p, c->d_first_state.p, c->d_last_state.p, c->d_winpart.p, c->d_win_off.p, c->d_fin.p, n_fin,
                       c->d_errflag.p, c->d_summary.p, c->n_win, c->extent, c->d_iv.p, cap, sq, sc, sm);
}

template <bool DEBUG> void launch_pileup(cl_ctx *c, const PileupArgs &a)
{
    const uint32_t grid = a.n_win8 * 8u;
    if (grid == 0) return;
    // (the ORF template parameter once selected a shorter threshold test for min_base_quality <= 128; one form
    // serves every threshold now and only ORF = true is instantiated)
#define CL_LAUNCH(DEEP_, LONG_) hipLaunchKernelGGL((k_pileup<(int)kT, DEBUG, true, DEEP_, LONG_>), dim3(grid), dim3(kBlock), 0, c->stream, a)
#define CL_LAUNCH_L(DEEP_) do { if (c->form == 2) CL_LAUNCH(DEEP_, 2); else CL_LAUNCH(DEEP_, 0); } while (0)
#define CL_LAUNCH_R(DEEP_, NP_) hipLaunchKernelGGL((k_pileup_rows<(int)kT, DEBUG, DEEP_, NP_, CL_ROWS_BLOCK>), dim3(grid), dim3(CL_ROWS_BLOCK), 0, c->stream, a)
#define CL_LAUNCH_RN(DEEP_) do { if (c->max_groups <= 63u) CL_LAUNCH_R(DEEP_, 8); else if (c->max_groups <= 16383u) CL_LAUNCH_R(DEEP_, 16); else CL_LAUNCH_R(DEEP_, 32); } while (0)
    // the 32-bit counter variant is used only when the window bounds asked for it (kNeedDeep)
    if (c->form == 3) {
        // pass-bit form: the counter planes by the deepest window's rows (4 per group): 8 planes count to 255
        if (!c->deep) CL_LAUNCH_RN(false); else CL_LAUNCH_RN(true);
    } else if (!c->deep) CL_LAUNCH_L(false); else CL_LAUNCH_L(true);
#undef CL_LAUNCH_RN
#undef CL_LAUNCH_R
#undef CL_LAUNCH_L
#undef CL_LAUNCH
}

cl_status enqueue(cl_ctx *c, bool debug, uint32_t *dbg_raw, uint32_t *dbg_qc, uint32_t *dbg_low)
{
    const bool prof = c->profiling && !debug;
    if (prof) {
        cl_status s = ensure_events(c);
        if (s != CL_OK) return s;
        if (c->ev_pending == cl_ctx::kEvSets) s = harvest_events(c);   // ring full: read back (host sync)
        if (s != CL_OK) return s;
    }
    hipEvent_t *ev = c->ev[c->ev_pending < cl_ctx::kEvSets ? c->ev_pending : 0];
    const Reads R = device_reads(c);

    // d_errflag is zero here: cleared at upload, and by the summary workgroup at the end of every run
    // (a run has no per-read kernel: the read ends and CIGAR checkpoints the long-read forms need are an index the host
    // built at upload; CL_K_PREP stays in the timing table as an empty slot)
    if (prof) HIP_TRY(c, hipEventRecord(ev[0], c->stream));
    PileupArgs a;
    a.R = R; a.o = c->dopt;
    a.rec = c->d_rec.p; a.heads = c->d_heads.p; a.end = c->d_end.p; a.win = c->d_win.p;
    a.wide_idx = c->d_wide_idx.p;
    a.ref = c->d_ref.p; a.refn = c->d_refn.p; a.lut = c->d_lut.p; a.state = c->d_state.p; a.winpart = c->d_winpart.p;
    a.runs = c->d_runs.p; a.first_state = c->d_first_state.p; a.last_state = c->d_last_state.p;
    if (debug) { HIP_TRY(c, c->d_state.reserve((size_t)c->n_win * kT + 16)); a.state = c->d_state.p; }
    a.extent = c->extent; a.n_win = c->n_win; a.n_win8 = (c->n_win + 7) / 8;
    a.dbg_raw = dbg_raw; a.dbg_qc = dbg_qc; a.dbg_low = dbg_low;
    a.win_wide = c->d_win_wide.p; a.err_flag = c->d_errflag.p;
    a.upl = (c->n_reads && c->n_qual <= 128ull * c->n_reads) ? 2u : 3u;   // by the mean read length
    a.ablate = c->tune_ablate;        // 0 outside tuning builds
    a.runtab = c->d_runtab.p; a.rows = c->d_rows.p;
    if (debug) launch_pileup<true>(c, a); else launch_pileup<false>(c, a);
    if (prof) HIP_TRY(c, hipEventRecord(ev[3], c->stream));
    launch_tail(c);
    if (prof) {
        HIP_TRY(c, hipEventRecord(ev[4], c->stream));
        c->ev_pending += 1;
    }
    HIP_TRY(c, hipGetLastError());
    return CL_OK;
}

} // namespace

extern "C" {

int cl_abi_version(void) { return CL_ABI_VERSION; }

int cl_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

cl_status cl_create(const cl_options *opt, int device_id, void *stream, cl_ctx **out)
{
    if (!opt || !out) return CL_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n)
        return CL_ERR_DEVICE;
    cl_ctx *c = new (std::nothrow) cl_ctx();
    if (!c) return CL_ERR_NOMEM;
    c->device = device_id;
    c->opt = *opt;
    // DUT_QUAL_FORM=bytes: the quality bytes themselves go to the device and k_pileup tests them there (the byte forms:
    // records for short reads, the run table for long ones) -- the forms of rounds 1-3, kept selectable so that their
    // measurements stay reproducible.  Default: the pass-bit form (k_pileup_rows).  Read per context.
    { const char *qf = getenv("DUT_QUAL_FORM"); c->bits = !(qf && strcmp(qf, "bytes") == 0); }
    // DUT_HEAD_SPAN (a test hook): spans beyond this many positions are cut into several heads -- 2^31 - 1 in earnest, which
    // only a contig of more than 2 Gb can hold; the tests put the seams into ordinary reads
    { const char *hs = getenv("DUT_HEAD_SPAN"); if (hs) { const unsigned long long v = strtoull(hs, nullptr, 0); if (v >= 1 && v <= kHeadSpanMax) c->head_span = (uint32_t)v; } }
    if (hipSetDevice(device_id) != hipSuccess) { delete c; return CL_ERR_DEVICE; }
    if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
    else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return CL_ERR_DEVICE; }
        c->own_stream = true;
    }
    Opts &o = c->dopt;
    o.min_depth = opt->min_depth; o.max_depth = opt->max_depth; o.min_mapq = opt->min_mapping_quality;
    o.min_depth_for_low_mapq = opt->min_depth_for_low_mapq; o.max_low_mapq = opt->max_low_mapq;
    o.max_low_mapq_fraction = opt->max_low_mapq_fraction;
    // pass_bytes: per byte (x + k + c) >> 1 has bit 7 set iff x >= min_base_quality
    o.ge_k = (opt->min_base_quality == 0 ? 255u : 256u - opt->min_base_quality) * 0x01010101u;
    o.ge_c = opt->min_base_quality == 0 ? 0x01010101u : 0u;
    o.md_all = opt->min_depth > 255 ? 1u : 0u;
    make_ge_consts((uint8_t)(opt->min_depth > 255 ? 255 : opt->min_depth), o.md_add, o.md_or, o.md_and);
    o.xd_on = (opt->max_depth >= 1 && opt->max_depth <= 254) ? 1u : 0u;
    make_ge_consts((uint8_t)(o.xd_on ? opt->max_depth + 1 : 255), o.xd_add, o.xd_or, o.xd_and);
#ifdef CL_TUNING
    if (const char *ab = getenv("CL_ABLATE")) {   // timing experiments: skips phases of k_pileup (results are then wrong)
        c->tune_ablate = (uint32_t)strtoul(ab, nullptr, 0);
        if (c->tune_ablate) fprintf(stderr, "[callable_loci] CL_ABLATE=%u: kernel phases are skipped, RESULTS ARE WRONG (tuning build)\n", c->tune_ablate);
    }
#endif
    std::vector<uint32_t> lut;
    build_lut(opt->max_low_mapq_fraction, lut);
    bool ok = c->d_lut.reserve(kLutSize) == hipSuccess &&
              c->d_summary.reserve(1) == hipSuccess && c->d_errflag.reserve(2) == hipSuccess &&
              hipMemcpy(c->d_lut.p, lut.data(), kLutSize * sizeof(uint32_t), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok || ensure_pins(c) != CL_OK) { cl_destroy(c); return CL_ERR_DEVICE; }
    *out = c;
    return CL_OK;
}

void cl_destroy(cl_ctx *c)
{
    if (!c) return;
    if (c->host_only) { delete c; return; }
    (void)hipSetDevice(c->device);
    StageTimer tmr;
    drop_prefetch(c);                                     // its copiers write into d_qual: joined before anything is released
    join_prealloc(c);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    tmr.lap("destroy: sync");
    c->d_pos.release(); c->d_mapq.release();
    c->d_qual.release(); c->d_rows.release(); c->d_ref.release(); c->d_end.release(); c->d_rec.release(); c->d_heads.release(); c->d_refn.release();
    c->d_win.release(); c->d_win_off.release(); c->d_state.release();
    c->d_wide_idx.release();
    c->d_runs.release(); c->d_first_state.release(); c->d_last_state.release(); c->d_win_wide.release();
    c->d_winpart.release(); c->d_lut.release(); c->d_summary.release();
    c->d_iv.release(); c->d_dbg.release(); c->d_fin.release(); c->d_errflag.release(); c->d_runtab.release(); c->site.release();
    for (int i = 0; i < 2; ++i) if (c->site_ev[i]) (void)hipEventDestroy(c->site_ev[i]);
    if (c->ev_made)
        for (int s = 0; s < cl_ctx::kEvSets; ++s)
            for (int i = 0; i <= CL_K_COUNT; ++i) (void)hipEventDestroy(c->ev[s][i]);
    tmr.lap("destroy: device buffers");
    c->ring.reset();
    tmr.lap("destroy: staging ring");
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    tmr.lap("destroy: stream, context");
}

const char *cl_last_error(const cl_ctx *c) { return c ? c->err.c_str() : "null context"; }

static cl_status cl_contig_begin_impl(cl_ctx *c, int32_t tid, uint32_t contig_len, const uint8_t *ref_bases,
                          uint64_t ref_len)
{
    if (!c) return CL_ERR_INVALID;
    if (contig_len > 0xFFF00000u) return fail(c, CL_ERR_RANGE, "contig length beyond the engine's 32-bit range");
    if (ref_len && !ref_bases) return fail(c, CL_ERR_INVALID, "ref_bases is null");
    if (!c->host_only) { drop_prefetch(c); join_prealloc(c); }
    take_staging(c);
    c->tid = tid; c->contig_len = contig_len;
    const uint64_t nref = std::min<uint64_t>(ref_len, contig_len);
    c->h_ref.clear(); c->h_ref.append(ref_bases, nref);
    c->h_pos.clear(); c->h_mapq.clear(); c->h_cigar.clear(); c->h_qual.clear();
    c->h_cigar_off.clear(); c->h_cigar_off.push_back(0u); c->h_qual_off.clear(); c->h_qual_off.push_back(0ull);
    c->h_iv.clear();
    c->q_dev = 0;
    c->h_wide_idx.clear(); c->h_wide_pos.clear(); c->span_n = 0; c->span_w = 0; c->n_wide = 0; c->host_max_end = 0;
    c->h_end.clear(); c->h_ck_x.clear(); c->h_ck_y.clear(); c->n_long = 0; c->host_err = 0; c->bounds_err = 0;
    c->h_rec_cnt.clear(); c->rec_counted = true;
    c->h_qbits.clear(); c->h_rb_off.clear(); c->h_sc_off.clear(); c->h_sc.clear(); c->host_sum_q = 0; c->host_n_ops = 0;
    c->host_sum_cov = 0; c->host_sum_mapq = 0;
    c->in_contig = true; c->uploaded = false; c->ran = false; c->has_long = false;
    return CL_OK;
}

cl_status cl_contig_begin(cl_ctx *c, int32_t tid, uint32_t contig_len, const uint8_t *ref_bases,
                          uint64_t ref_len)
{
    // no exception leaves the library through the C ABI
    try { return cl_contig_begin_impl(c, tid, contig_len, ref_bases, ref_len); }
    catch (const std::bad_alloc &) { return fail(c, CL_ERR_NOMEM, "out of memory"); }
    catch (...) { return fail(c, CL_ERR_INVALID, "internal error"); }
}


namespace {
// quality bytes staged on the host so far go to the device, behind the ones already there
cl_status flush_staged_qual(cl_ctx *c)
{
    if (c->h_qual.empty()) return CL_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, c->d_qual.grow_keep(c->q_dev + c->h_qual.size() + 2 * kQualPad, kQualPad + c->q_dev, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_qual.p + kQualPad + c->q_dev, c->h_qual.data(), c->h_qual.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->q_dev += c->h_qual.size();
    c->h_qual.clear();
    return CL_OK;
}
constexpr uint64_t kDirectQual = 4u << 20;   // tiles with at least this many quality bytes skip the host staging copy
} // namespace

cl_status cl_contig_prefetch_qual(cl_ctx *c, const uint8_t *qual, uint64_t n_bytes)
{
    if (!c || !c->in_contig || c->uploaded) return fail(c, CL_ERR_INVALID, "cl_contig_prefetch_qual outside cl_contig_begin .. upload");
    if (c->bits) return CL_OK;                                  // pass-bit form: no quality byte goes to the device
    if (!qual || n_bytes < kDirectQual) return CL_OK;            // small tiles are staged on the host anyway
    try {
        drop_prefetch(c);
        cl_status fs = flush_staged_qual(c);
        if (fs != CL_OK) return fs;
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, c->d_qual.grow_keep(c->q_dev + n_bytes + 2 * kQualPad, c->q_dev ? kQualPad + c->q_dev : 0, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        cl_status rs = ring_start(c, c->d_qual.p + kQualPad + c->q_dev, n_bytes,
                                  [qual](uint64_t off, uint64_t len, uint8_t *out) { memcpy(out, qual + off, len); });
        if (rs != CL_OK) { (void)ring_finish(c); return rs; }
        c->pf_src = qual; c->pf_n = n_bytes; c->pf_off = c->q_dev; c->pf_active = true;
        return CL_OK;
    }
    catch (const std::bad_alloc &) { return fail(c, CL_ERR_NOMEM, "out of memory"); }
    catch (...) { return fail(c, CL_ERR_INVALID, "internal error"); }
}

// Pass-bit form: what cl_contig_upload will need on the device is known from the contig's length and the caller's hint --
// allocated on a thread of its own while the caller admits and pushes the reads (a fresh context spends 15-20 ms of a
// chr21-sized contig's first pass in hipMalloc otherwise).  Only sizes are guessed here: cl_contig_upload reserves what it
// needs again and so makes up for a guess that fell short or an allocation that failed.  Joined before any buffer is used.
static void start_prealloc(cl_ctx *c, uint64_t n_reads, uint64_t n_qual)
{
    join_prealloc(c);
    const uint32_t contig_len = c->contig_len;
    {   // nothing to do for a context whose buffers hold this contig already (the usual case from its second contig on):
        // no thread is made for that
        const size_t n_win = ((size_t)contig_len + kT - 1) / kT + 1, padded = n_win * kT;
        const size_t rows_want = n_qual ? ((n_qual / kT) * 17 / 40 + n_win + 1024) * (dut::kRowGroupWords / 4) : 0;
        const bool enough = c->d_win.cap >= n_win + 1 && c->d_win_off.cap >= n_win + 1 && c->d_winpart.cap >= n_win + 1 &&
                            c->d_fin.cap >= n_win / kFinBlock + 2 && c->d_runs.cap >= padded + 16 && c->d_first_state.cap >= n_win + 1 &&
                            c->d_last_state.cap >= n_win + 1 && c->d_win_wide.cap >= n_win + 1 && c->d_refn.cap >= padded / 32 + 4 &&
                            c->d_heads.cap >= (n_reads ? (size_t)n_reads + 1 : 0) && c->d_rows.cap >= rows_want && c->d_iv.cap != 0;
        if (enough) return;
    }
    c->prealloc = dut::spawn_or_run([c, n_reads, n_qual, contig_len]() {
        if (hipSetDevice(c->device) != hipSuccess) return;
        const size_t n_win = ((size_t)contig_len + kT - 1) / kT + 1, padded = n_win * kT;
        (void)c->d_win.reserve(n_win + 1); (void)c->d_win_off.reserve(n_win + 1); (void)c->d_winpart.reserve(n_win + 1);
        (void)c->d_fin.reserve(n_win / kFinBlock + 2); (void)c->d_runs.reserve(padded + 16);
        (void)c->d_first_state.reserve(n_win + 1); (void)c->d_last_state.reserve(n_win + 1); (void)c->d_win_wide.reserve(n_win + 1);
        (void)c->d_refn.reserve(padded / 32 + 4);
        if (n_reads) (void)c->d_heads.reserve((size_t)n_reads + 1);
        if (n_qual) (void)c->d_rows.reserve(((n_qual / kT) * 17 / 40 + n_win + 1024) * (dut::kRowGroupWords / 4));   // stream_rows' own estimate
        if (c->d_iv.cap == 0) (void)c->d_iv.reserve(1u << 20);
    });
}

cl_status cl_contig_reserve(cl_ctx *c, uint64_t n_reads, uint64_t n_cigar_ops, uint64_t n_qual_bytes)
{
    if (!c || !c->in_contig || c->uploaded) return fail(c, CL_ERR_INVALID, "cl_contig_reserve outside cl_contig_begin .. upload");
    drop_prefetch(c);                                        // the quality buffer may move below
    try {
        c->h_pos.reserve(n_reads); c->h_mapq.reserve(n_reads);
        c->h_cigar_off.reserve(n_reads + 1); c->h_qual_off.reserve(n_reads + 1);
        c->h_cigar.reserve(n_cigar_ops);
        if (c->bits) c->h_qbits.reserve(((n_qual_bytes + 63) >> 6) + 2);
    } catch (const std::bad_alloc &) {
        return fail(c, CL_ERR_NOMEM, "host staging allocation failed");
    }
    if (c->bits) {
        if (!c->host_only && c->h_pos.empty()) start_prealloc(c, n_reads, n_qual_bytes);
        return CL_OK;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, c->d_qual.grow_keep(n_qual_bytes + 2 * kQualPad, c->q_dev ? kQualPad + c->q_dev : 0, c->stream));
    return CL_OK;
}

// cl_push_reads of the pass-bit form: nothing goes to the device here.  One walk over the tile, in chunks on all host
// threads, does everything that needs the reads' CIGAR operations and quality bytes while both are in the cache:
// validation (what protects the later walks' indexing; the CIGAR shapes htslib's resolve_cigar2 asserts on), every
// read's end (pos + bam_cigar2rlen, the pileup node span), the base-quality test of mod.rs:33 -- one
// bit per base, qual_pack.cpp --, the read's share of summed_baseq (contig_profiler.rs:65-70) and, for every read with
// mapq >= min_mapping_quality that is not a plain match, its pass bits mapped through the CIGAR into REFERENCE order
// (pass_rows.h): what cl_contig_upload then lays out as rows is a string of bits per read, and no CIGAR is staged at all
// (but those of reads whose span dwarfs their query: long N gaps).  A refused tile leaves the context as it was.
// n bits of `src` from bit offset sb into dst[0, ceil(n / 64)): whole words, zeros above bit n
static inline void copy_bits_to_words(uint64_t *dst, const uint64_t *src, unsigned long long sb, unsigned long long n)
{
    const unsigned long long w0 = sb >> 6;
    const uint32_t s = (uint32_t)(sb & 63ull);
    const unsigned long long nw = (n + 63) >> 6;
    for (unsigned long long w = 0; w < nw; ++w) {
        const unsigned long long left = n - (w << 6);                    // bits still wanted from here on (>= 1)
        const uint32_t want = left >= 64ull ? 64u : (uint32_t)left;
        uint64_t v = src[w0 + w] >> s;
        if (s + want > 64u) v |= src[w0 + w + 1] << (64u - s);
        if (want < 64u) v &= (1ull << want) - 1ull;
        dst[w] = v;
    }
}

// (pbits / psum: the packed variant, cl_push_reads_bits -- the caller has taken the base-quality test: bit
// t->qual_off[i] + k of pbits <-> quality value k of read i, psum[i] the read's share of summed_baseq; t->qual is unused)
static cl_status push_reads_bits(cl_ctx *c, const cl_read_tile *t, uint32_t cig0, uint64_t q0, uint64_t ncig, uint64_t nq,
                                 const uint64_t *pbits = nullptr, const uint32_t *psum = nullptr)
{
    const uint64_t n = t->n_reads;
    const uint64_t rbase = c->h_pos.size();
    StageTimer tmr;
    const size_t grain = dut::grain_for(n, 65536);
    const size_t nchunk = (n + grain - 1) / grain;
    struct Chunk {
        int bad = 0; uint32_t err = 0, span_n = 0, span_w = 0; uint64_t max_end = 0, sum_q = 0, n_ops = 0, n_words = 0, n_sc = 0;
        uint64_t sum_cov = 0, sum_mapq = 0;
        std::vector<uint32_t> wide;
    };
    std::vector<Chunk> ch(nchunk);
    const int32_t last0 = c->h_pos.empty() ? 0 : c->h_pos.back();
    try {
        c->h_rec_cnt.reserve(rbase + n); c->h_end.reserve(rbase + n);
        c->h_rb_off.reserve(rbase + n + 1); c->h_sc_off.reserve(rbase + n + 1);
    } catch (const std::bad_alloc &) {
        return fail(c, CL_ERR_NOMEM, "host staging allocation failed");
    }
    // (entries [rbase, rbase + n) of the staging arrays are written below; their sizes follow when the tile is accepted,
    // so a refused tile leaves nothing but unused capacity behind)
    uint32_t *const h_end = c->h_end.data() + rbase, *const h_rec_cnt = c->h_rec_cnt.data() + rbase;
    unsigned long long *const rb_off = c->h_rb_off.data() + rbase;      // first the reads' word counts (| kRowSparse), then their offsets
    uint32_t *const sc_off = c->h_sc_off.data() + rbase;
    const uint32_t min_mapq = c->opt.min_mapping_quality;
    const uint8_t min_bq = c->opt.min_base_quality;
    const uint64_t head_span = c->head_span;
    const int plevel = dut::qual_pack_level();
    const uint8_t *const qsrc = t->qual ? t->qual + q0 : nullptr;
    // ---- first: what needs the CIGAR operations only -- validation, ends, how many words of bits every read will leave ----
    dut::parallel_for(nchunk, 1, [&](size_t k) {
        Chunk &o = ch[k];
        const size_t a = k * grain, b = std::min<size_t>(n, a + grain);
        int32_t last = a ? t->pos[a - 1] : last0;
        for (size_t i = a; i < b; ++i) {
            const int32_t p = t->pos[i];
            if (p < 0 || (uint32_t)p >= c->contig_len) { if (!o.bad) o.bad = 1; }
            else if (p < last) { if (!o.bad) o.bad = 2; }
            last = p;
            h_end[i] = (uint32_t)p; h_rec_cnt[i] = 0u; rb_off[i] = 0ull; sc_off[i] = 0u;
            if (t->cigar_off[i + 1] < t->cigar_off[i] || t->qual_off[i + 1] < t->qual_off[i]) { if (!o.bad) o.bad = 3; continue; }
            if (t->cigar_off[i] < cig0 || t->cigar_off[i + 1] > cig0 + ncig) { if (!o.bad) o.bad = 3; continue; }
            if (t->qual_off[i] < q0 || t->qual_off[i + 1] > q0 + nq) { if (!o.bad) o.bad = 3; continue; }
            const uint32_t q0i = t->cigar_off[i], nops = t->cigar_off[i + 1] - q0i;
            const uint32_t *cig = t->cigar + q0i;
            o.n_ops += nops;
            unsigned long long l = 0;
            uint32_t zero_len = 0;
            for (uint32_t q = 0; q < nops; ++q) {
                const uint32_t cw = cig[q], len = cw >> 4;
                const uint32_t radv = (0x18Du >> (cw & 15u)) & 1u;              // M D N = X consume the reference
                l += len & (0u - radv);
                zero_len |= radv & (len == 0u ? 1u : 0u);                      // zero-length reference-consuming op
            }
            if (zero_len) o.err |= kErrCigar;
            // a read that reaches a column with a single non-match op is undefined in htslib
            if (l > 0 && nops == 1u && !(((0x181u >> (cig[0] & 15u)) & 1u) != 0u)) o.err |= kErrCigar;
            const uint32_t sp = l > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)l;
            // an end beyond the engine's 32-bit coordinate range: flagged; the read then spans nothing
            if (l <= 0xFFFF0000ull - (uint64_t)p) { h_end[i] = (uint32_t)((uint64_t)p + l); o.max_end = std::max<uint64_t>(o.max_end, (uint64_t)p + l); }
            else o.err |= kErrRange;
            if (sp > kWideSpan) { o.wide.push_back((uint32_t)i); o.span_w = std::max(o.span_w, sp); }
            else o.span_n = std::max(o.span_n, sp);
            const bool in_pileup = h_end[i] != (uint32_t)p;
            const uint64_t span = h_end[i] - (uint32_t)p;
            // the heads k_pileup_rows reads: one, unless the span is beyond what a head holds
            h_rec_cnt[i] = in_pileup ? (uint32_t)((span + head_span - 1) / head_span) : 0u;
            // the read's shares of summed_coverage and summed_mapq (contig_profiler.rs:79-82, :74 -- over its columns,
            // D and N included: SURVEY 8a-7)
            o.sum_cov += span;
            if (t->mapq[i] >= min_mapq) o.sum_mapq += (uint64_t)t->mapq[i] * span;
            const unsigned long long ql = t->qual_off[i + 1] - t->qual_off[i];
            if (!in_pileup || !ql || t->mapq[i] < min_mapq) continue;          // in no row (mod.rs:25, :33)
            uint64_t nw;
            if (nops == 1u) nw = (std::min<uint64_t>(span, ql) + 63) >> 6;     // a plain match: its thresholded string as it is
            else if (span > 4 * ql + 1024) {                                   // a span that dwarfs the query: query order + the CIGAR
                nw = ((ql + 63) >> 6) | dut::kRowSparse;
                sc_off[i] = nops; o.n_sc += nops;
            } else nw = (span + 63) >> 6;                                      // mapped into reference order below
            rb_off[i] = nw;
            o.n_words += nw & ~dut::kRowSparse;
        }
    });
    tmr.lap("push: validate + spans");
    // (entry rbase of the two offset arrays closes the tiles before this one; the walk above used it for this tile's
    // first read: put back whenever the tile is refused)
    const unsigned long long closing_rb = c->h_qbits.size();
    const uint32_t closing_sc = (uint32_t)c->h_sc.size();
    auto refuse = [&](cl_status st, const char *m) { rb_off[0] = closing_rb; sc_off[0] = closing_sc; return fail(c, st, m); };
    for (const Chunk &o : ch) {                                // the first offence in tile order decides the message
        if (o.bad == 1) return refuse(CL_ERR_INVALID, "read position outside [0, contig_len): the region fetch (mod.rs:53) never yields it");
        if (o.bad == 2) return refuse(CL_ERR_UNSORTED, "reads are not coordinate sorted");
        if (o.bad == 3) return refuse(CL_ERR_INVALID, "offset arrays must be non-decreasing");
    }
    // ---- where every chunk's strings go: behind the contig's ----
    std::vector<uint64_t> rb_base(nchunk + 1), sc_base(nchunk + 1);
    rb_base[0] = c->h_qbits.size(); sc_base[0] = c->h_sc.size();
    for (size_t k = 0; k < nchunk; ++k) { rb_base[k + 1] = rb_base[k] + ch[k].n_words; sc_base[k + 1] = sc_base[k] + ch[k].n_sc; }
    if (sc_base[nchunk] > 0xFFFFFFF0ull) return refuse(CL_ERR_RANGE, "more than 2^32 CIGAR operations of gapped reads in one contig");
    size_t n_wide_new = 0;
    for (const Chunk &o : ch) n_wide_new += o.wide.size();
    try {
        c->h_qbits.reserve(rb_base[nchunk] + 2); c->h_sc.reserve(sc_base[nchunk] + 1);
        c->h_pos.reserve(rbase + n); c->h_mapq.reserve(rbase + n);
        c->h_wide_idx.reserve(c->h_wide_idx.size() + n_wide_new); c->h_wide_pos.reserve(c->h_wide_pos.size() + n_wide_new);
    } catch (const std::bad_alloc &) {
        return refuse(CL_ERR_NOMEM, "host staging allocation failed");
    }
    uint64_t *const bits = c->h_qbits.data();
    uint32_t *const scw = c->h_sc.data();
    // ---- then: what needs the quality bytes -- the base-quality test (one bit per base), the reads' shares of
    //      summed_baseq, and the bits of every read that is not a plain match mapped through its CIGAR into reference
    //      order -- written straight to their place (the chunks' ranges are disjoint) ----
    std::atomic<bool> oom{false};
    dut::parallel_for(nchunk, 1, [&](size_t k) {
        Chunk &o = ch[k];
        const size_t a = k * grain, b = std::min<size_t>(n, a + grain);
        uint64_t wat = rb_base[k], sat = sc_base[k];
        RawVec<uint64_t> qw;                                    // a read's query-order bits
        RawVec<dut::QueryStretch> um;                           // where its inserted / clipped bases lie
        try {
        for (size_t i = a; i < b; ++i) {
            const unsigned long long cnt = rb_off[i];
            const uint64_t nw = cnt & ~dut::kRowSparse;
            rb_off[i] = wat | (cnt & dut::kRowSparse);
            const uint32_t nsc = sc_off[i];
            sc_off[i] = (uint32_t)sat;
            if (!nw) continue;
            const uint32_t q0i = t->cigar_off[i], nops = t->cigar_off[i + 1] - q0i;
            const uint32_t *cig = t->cigar + q0i;
            const unsigned long long ql = t->qual_off[i + 1] - t->qual_off[i];
            const uint8_t *q = qsrc ? qsrc + (t->qual_off[i] - q0) : nullptr;
            const uint64_t span = h_end[i] - (uint32_t)t->pos[i];
            if (pbits) {
                // the packed variant: the bits are there, in query order from bit qual_off[i]
                const bool sparse = (cnt & dut::kRowSparse) != 0ull;
                o.sum_q += psum[i];
                if (nops == 1u) copy_bits_to_words(bits + wat, pbits, t->qual_off[i], std::min<uint64_t>(span, ql));
                else if (sparse) {
                    copy_bits_to_words(bits + wat, pbits, t->qual_off[i], ql);
                    memcpy(scw + sat, cig, (size_t)nops * sizeof(uint32_t));
                } else {
                    const uint64_t nqw = (ql + 63) >> 6;
                    qw.resize(nqw + 2); qw[nqw] = 0ull; qw[nqw + 1] = 0ull;
                    copy_bits_to_words(qw.data(), pbits, t->qual_off[i], ql);
                    um.resize(nops + 1);
                    size_t n_um = 0; unsigned long long qlen = 0;
                    dut::ref_bits_from_query(qw.data(), ql, cig, nops, bits + wat, um.data(), &n_um, &qlen);
                }
            } else if (nops == 1u) {
                o.sum_q += dut::qual_pass_read(q, std::min<uint64_t>(span, ql), min_bq, bits + wat, plevel);
            } else {
                const uint64_t nqw = (ql + 63) >> 6;
                const bool sparse = (cnt & dut::kRowSparse) != 0ull;
                uint64_t *dstq;
                if (sparse) dstq = bits + wat;
                else { qw.resize(nqw + 2); dstq = qw.data(); dstq[nqw] = 0ull; dstq[nqw + 1] = 0ull; }   // (two readable words behind the bits: the mapping looks one word ahead of a clamped position)
                const uint64_t all = dut::qual_pass_read(q, ql, min_bq, dstq, plevel);      // every byte of the string that passes ...
                um.resize(nops + 1);
                size_t n_um = 0; unsigned long long qlen = 0;
                if (sparse) {
                    // (a gapped read keeps its query-order bits and its CIGAR; the mapping below runs into a scratch word
                    // count of zero: only the list of its inserted / clipped bases is wanted)
                    memcpy(scw + sat, cig, (size_t)nops * sizeof(uint32_t));
                    unsigned long long y = 0;
                    for (uint32_t j = 0; j < nops; ++j) {
                        const uint32_t cw = cig[j], op = cw & 15u, l = cw >> 4;
                        if (!((0x193u >> op) & 1u)) continue;
                        if (!((0x181u >> op) & 1u)) { um[n_um].y = y; um[n_um].l = l; ++n_um; }
                        y += l;
                    }
                    qlen = y;
                } else {
                    // ... mapped through the CIGAR into reference order, here where the operations are in the cache
                    dut::ref_bits_from_query(qw.data(), ql, cig, nops, bits + wat, um.data(), &n_um, &qlen);
                }
                o.sum_q += all - dut::unmatched_pass_sum(q, ql, um.data(), n_um, qlen, min_bq);   // ... minus those of inserted / clipped bases
            }
            wat += nw; sat += nsc;
        }
        } catch (const std::bad_alloc &) { oom.store(true); }
    });
    tmr.lap("push: pass bits in reference order");
    if (oom.load()) return refuse(CL_ERR_NOMEM, "host staging allocation failed");
    // ---- the tile is accepted (nothing below can fail: the capacity is there) ----
    for (const Chunk &o : ch) {
        c->host_err |= o.err; c->host_sum_q += o.sum_q; c->host_n_ops += o.n_ops;
        c->host_sum_cov += o.sum_cov; c->host_sum_mapq += o.sum_mapq;
        c->span_n = std::max(c->span_n, o.span_n); c->span_w = std::max(c->span_w, o.span_w);
        c->host_max_end = std::max(c->host_max_end, o.max_end);
        for (uint32_t i : o.wide) { c->h_wide_idx.push_back((uint32_t)(rbase + i)); c->h_wide_pos.push_back(t->pos[i]); }
    }
    c->h_qbits.resize(rb_base[nchunk]); c->h_sc.resize(sc_base[nchunk]);
    c->h_qbits.data()[rb_base[nchunk]] = 0ull;                 // the word deposit_bits may read behind the last string
    c->h_end.resize(rbase + n); c->h_rec_cnt.resize(rbase + n);
    c->h_rb_off.resize(rbase + n + 1); c->h_sc_off.resize(rbase + n + 1);
    rb_off[n] = rb_base[nchunk]; sc_off[n] = (uint32_t)sc_base[nchunk];
    c->h_pos.append(t->pos, n);
    c->h_mapq.append(t->mapq, n);
    tmr.lap("push: stage small arrays");
    c->q_dev += nq;                                          // (the contig's quality bytes so far: none on the device)
    return CL_OK;
}

static cl_status cl_push_reads_impl(cl_ctx *c, const cl_read_tile *t)
{
    if (!c || !t) return CL_ERR_INVALID;
    if (!c->in_contig || c->uploaded) return fail(c, CL_ERR_INVALID, "cl_push_reads outside cl_contig_begin .. upload");
    const uint64_t n = t->n_reads;
    if (n == 0) return CL_OK;
    if (!t->pos || !t->mapq || !t->cigar_off || !t->qual_off) return fail(c, CL_ERR_INVALID, "null tile array");
    if (c->h_pos.size() + n >= (1ull << 29)) return fail(c, CL_ERR_RANGE, "more than 2^29 reads in one contig");
    const uint32_t cig0 = t->cigar_off[0];
    const uint64_t q0 = t->qual_off[0];
    if (t->cigar_off[n] < cig0 || t->qual_off[n] < q0) return fail(c, CL_ERR_INVALID, "offset arrays must be non-decreasing");
    const uint64_t ncig = (uint64_t)t->cigar_off[n] - cig0;
    const uint64_t nq = t->qual_off[n] - q0;
    if (ncig && !t->cigar) return fail(c, CL_ERR_INVALID, "null cigar array");
    if (nq && !t->qual) return fail(c, CL_ERR_INVALID, "null qual array");
    if (c->h_cigar.size() + ncig > 0xFFFFFFF0ull) return fail(c, CL_ERR_RANGE, "more than 2^32 CIGAR operations in one contig");
    if (c->q_dev + c->h_qual.size() + nq >= (1ull << 38)) return fail(c, CL_ERR_RANGE, "more than 2^38 quality bytes in one contig");
    if (c->bits) return push_reads_bits(c, t, cig0, q0, ncig, nq);      // the pass-bit form: no quality byte goes to the device
    const uint32_t cbase = (uint32_t)c->h_cigar.size();
    const uint64_t rbase = c->h_pos.size();

    // ---- a large tile: its quality bytes go from the caller's buffer to the device through the pinned staging
    //      ring, and they start now (unless cl_contig_prefetch_qual already sent exactly these bytes): the copier
    //      threads fill pinned buffers and queue their transfers while this thread validates the tile and stages
    //      the small arrays.  All are joined before the call returns (the caller's buffer is free again then);
    //      nothing of the context changes if the tile turns out to be invalid. ----
    struct RingGuard { cl_ctx *c; bool active = false; ~RingGuard() { if (active) (void)ring_finish(c); } } ring{c};
    const bool direct = nq >= kDirectQual;
    if (direct) {
        const uint8_t *src = t->qual + q0;
        const bool prefetched = c->pf_active && c->pf_src == src && c->pf_n == nq && c->h_qual.empty() && c->pf_off == c->q_dev;
        if (prefetched) {
            c->pf_active = false; c->pf_src = nullptr; c->pf_n = 0;
            ring.active = true;                               // the transfer in flight is this tile's
        } else {
            drop_prefetch(c);
            cl_status fs = flush_staged_qual(c);
            if (fs != CL_OK) return fs;
            HIP_TRY(c, hipSetDevice(c->device));
            HIP_TRY(c, c->d_qual.grow_keep(c->q_dev + nq + 2 * kQualPad, c->q_dev ? kQualPad + c->q_dev : 0, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));          // the (re)allocation above is done
            cl_status rs = ring_start(c, c->d_qual.p + kQualPad + c->q_dev, nq,
                                      [src](uint64_t off, uint64_t len, uint8_t *out) { memcpy(out, src + off, len); });
            if (rs != CL_OK) return rs;
            ring.active = true;
        }
    } else if (c->pf_active) {
        drop_prefetch(c);
    }
    const unsigned long long qbase = c->q_dev + c->h_qual.size();
    StageTimer tmr;

    // ---- the one walk over every CIGAR of the tile: validation that protects the kernels' indexing; the end of every
    //      read (pos + bam_cigar2rlen, the pileup node span) -- the longest ordinary span bounds every
    //      window's candidate range, reads wider than kWideSpan get their own list; the CIGAR shapes htslib's
    //      resolve_cigar2 asserts on or indexes out of bounds for; and, for reads with more than kLongOps operations,
    //      the (reference, query) position before every operation whose index in the contig's CIGAR array is a multiple
    //      of 64, where k_pileup starts its walk of such a read instead of at its first operation.  In chunks, on all
    //      host threads. ----
    // (chunks of at most 65 536 reads; long reads -- few records, thousands of operations each -- get smaller ones so
    // that every thread has some)
    const size_t grain = dut::grain_for(n, 65536);
    const size_t nchunk = (n + grain - 1) / grain;
    struct Chunk { int bad = 0; uint32_t n_long = 0, err = 0; uint32_t span_n = 0, span_w = 0; uint64_t max_end = 0; std::vector<uint32_t> wide; };
    std::vector<Chunk> ch(nchunk);
    const int32_t last0 = c->h_pos.empty() ? 0 : c->h_pos.back();
    try {
        c->h_rec_cnt.reserve(rbase + n);
        c->h_end.reserve(rbase + n);                           // entries [rbase, rbase + n) are written below; the
        c->h_ck_x.reserve(((cbase + ncig) >> 6) + 2);          // sizes follow when the tile is accepted (a refused
        c->h_ck_y.reserve(((cbase + ncig) >> 6) + 2);          // tile leaves only unused capacity behind)
    } catch (const std::bad_alloc &) {
        return fail(c, CL_ERR_NOMEM, "host staging allocation failed");
    }
    uint32_t *const h_end = c->h_end.data() + rbase, *const h_ck_x = c->h_ck_x.data(), *const h_ck_y = c->h_ck_y.data();
    uint32_t *const h_rec_cnt = c->h_rec_cnt.data() + rbase;
    const uint32_t min_mapq = c->opt.min_mapping_quality;
    // (a tile of long-read shape -- 8 or more operations per read -- will not get the short-read form: its reads'
    // records are not counted here, that would be a second pass over every operation)
    const bool count_recs = ncig < 8ull * n;
    dut::parallel_for(nchunk, 1, [&](size_t k) {
        Chunk &o = ch[k];
        const size_t a = k * grain, b = std::min<size_t>(n, a + grain);
        int32_t last = a ? t->pos[a - 1] : last0;
        for (size_t i = a; i < b; ++i) {
            const int32_t p = t->pos[i];
            if (p < 0 || (uint32_t)p >= c->contig_len) { if (!o.bad) o.bad = 1; }
            else if (p < last) { if (!o.bad) o.bad = 2; }
            last = p;
            h_end[i] = (uint32_t)p; h_rec_cnt[i] = 0u;
            if (t->cigar_off[i + 1] < t->cigar_off[i] || t->qual_off[i + 1] < t->qual_off[i]) { if (!o.bad) o.bad = 3; continue; }
            if (t->cigar_off[i] < cig0 || t->cigar_off[i + 1] > cig0 + ncig) { if (!o.bad) o.bad = 3; continue; }
            if (t->qual_off[i] < q0 || t->qual_off[i + 1] > q0 + nq) { if (!o.bad) o.bad = 3; continue; }
            const uint32_t q0i = t->cigar_off[i], q1i = t->cigar_off[i + 1], nops = q1i - q0i;
            unsigned long long l = 0;
            if (nops <= kLongOps) {
                for (uint32_t q = q0i; q < q1i; ++q) {
                    const uint32_t cw = t->cigar[q], len = cw >> 4;
                    const bool radv = ((0x18Du >> (cw & 15u)) & 1u) != 0u;      // M D N = X consume the reference
                    l += radv ? len : 0u;
                    if (radv && len == 0u) o.err |= kErrCigar;                 // zero-length reference-consuming op
                }
                // a read that reaches a column with a single non-match op is undefined in htslib
                if (l > 0 && nops == 1u && !(((0x181u >> (t->cigar[q0i] & 15u)) & 1u) != 0u)) o.err |= kErrCigar;
            } else {
                o.n_long += 1;
                uint32_t yq = 0;                                               // query advance (M I S = X), modulo 2^32
                const uint32_t shift = cbase - cig0;                           // tile op index -> contig op index (mod 2^32)
                for (uint32_t q = q0i; q < q1i; ++q) {
                    const uint32_t kc = q + shift;
                    if ((kc & 63u) == 0u) {
                        const unsigned long long cx = (unsigned long long)(uint32_t)p + l;
                        h_ck_x[kc >> 6] = cx > 0xFFFF0000ull ? 0xFFFF0000u : (uint32_t)cx;
                        h_ck_y[kc >> 6] = yq;
                    }
                    const uint32_t cw = t->cigar[q], len = cw >> 4, op = cw & 15u;
                    const bool radv = ((0x18Du >> op) & 1u) != 0u, qadv = ((0x193u >> op) & 1u) != 0u;
                    l += radv ? len : 0u;
                    yq += qadv ? len : 0u;
                    if (radv && len == 0u) o.err |= kErrCigar;
                }
            }
            const uint32_t sp = l > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)l;
            // an end beyond the engine's 32-bit coordinate range: flagged; the read then spans nothing
            if (l <= 0xFFFF0000ull - (uint64_t)p) { h_end[i] = (uint32_t)((uint64_t)p + l); o.max_end = std::max<uint64_t>(o.max_end, (uint64_t)p + l); }
            else o.err |= kErrRange;
            if (sp > kWideSpan) { o.wide.push_back((uint32_t)i); o.span_w = std::max(o.span_w, sp); }
            else o.span_n = std::max(o.span_n, sp);
            // the records the short-read form would get for this read (counted here, where its CIGAR is hot)
            if (count_recs) {
                const unsigned long long ql = t->qual_off[i + 1] - t->qual_off[i];
                const uint32_t mq = t->mapq[i];
                uint32_t cnt;
                // one M/=/X operation as long as the qualities (96 reads in 100 of aligner output): one record, no walk
                if (nops == 1u && h_end[i] != (uint32_t)p && ql < 0x10000ull && ((0x181u >> (t->cigar[q0i] & 15u)) & 1u) && (t->cigar[q0i] >> 4) == ql)
                    cnt = 1u;
                else cnt = gen_read_recs(p, h_end[i], mq, min_mapq, t->cigar + q0i, nops, 0ull, ql, [](uint32_t, const ReadRec &) {});
                h_rec_cnt[i] = cnt;
            }
        }
    });
    tmr.lap("push: validate + spans");
    for (const Chunk &o : ch) {                                // the first offence in tile order decides the message
        if (o.bad == 1) return fail(c, CL_ERR_INVALID, "read position outside [0, contig_len): the region fetch (mod.rs:53) never yields it");
        if (o.bad == 2) return fail(c, CL_ERR_UNSORTED, "reads are not coordinate sorted");
        if (o.bad == 3) return fail(c, CL_ERR_INVALID, "offset arrays must be non-decreasing");
    }

    // ---- staging of the small arrays (offsets rebased onto the contig's); undone if anything below fails, so that
    //      a refused tile leaves the context as it was ----
    struct Undo {
        cl_ctx *c; size_t n_pos, n_cig, n_qual, n_wide; uint32_t n_long, host_err; bool has_long; uint32_t span_n, span_w; uint64_t max_end; bool armed = true;
        ~Undo()
        {
            if (!armed) return;
            c->h_pos.resize(n_pos); c->h_mapq.resize(n_pos); c->h_cigar.resize(n_cig); c->h_qual.resize(n_qual);
            c->h_cigar_off.resize(n_pos + 1); c->h_qual_off.resize(n_pos + 1);
            c->h_wide_idx.resize(n_wide); c->h_wide_pos.resize(n_wide); c->n_long = n_long; c->host_err = host_err;
            c->h_end.resize(n_pos); c->h_ck_x.resize((n_cig >> 6) + 2); c->h_ck_y.resize((n_cig >> 6) + 2);
            c->h_rec_cnt.resize(n_pos);
            c->has_long = has_long; c->span_n = span_n; c->span_w = span_w; c->host_max_end = max_end;
        }
    } undo{c, c->h_pos.size(), c->h_cigar.size(), c->h_qual.size(), c->h_wide_idx.size(), c->n_long, c->host_err, c->has_long, c->span_n, c->span_w, c->host_max_end};
    try {
        for (const Chunk &o : ch) {
            if (o.n_long) c->has_long = true;
            c->n_long += o.n_long; c->host_err |= o.err;
            c->span_n = std::max(c->span_n, o.span_n); c->span_w = std::max(c->span_w, o.span_w);
            c->host_max_end = std::max(c->host_max_end, o.max_end);
            for (uint32_t i : o.wide) { c->h_wide_idx.push_back((uint32_t)(rbase + i)); c->h_wide_pos.push_back(t->pos[i]); }
        }
        c->h_end.resize(rbase + n); c->h_ck_x.resize(((cbase + ncig) >> 6) + 2); c->h_ck_y.resize(((cbase + ncig) >> 6) + 2);
        c->h_rec_cnt.resize(rbase + n);
        if (!count_recs) c->rec_counted = false;
        c->h_pos.append(t->pos, n);
        c->h_mapq.append(t->mapq, n);
        c->h_cigar.append(t->cigar + cig0, ncig);
        if (!direct) c->h_qual.insert(c->h_qual.end(), t->qual + q0, t->qual + q0 + nq);
        const size_t o0 = c->h_cigar_off.size();               // == rbase + 1: entry r+1 closes read r
        c->h_cigar_off.resize(o0 + n);
        c->h_qual_off.resize(o0 + n);
        uint32_t *co = c->h_cigar_off.data() + o0 - 1;
        unsigned long long *qo = c->h_qual_off.data() + o0 - 1;
        dut::parallel_for(n, 262144, [&](size_t i) {
            co[i + 1] = cbase + (t->cigar_off[i + 1] - cig0);
            qo[i + 1] = qbase + (t->qual_off[i + 1] - q0);
        });
    } catch (const std::bad_alloc &) {
        return fail(c, CL_ERR_NOMEM, "host staging allocation failed");
    }
    tmr.lap("push: stage small arrays");
    if (ring.active) {
        ring.active = false;
        cl_status rs = ring_finish(c);                         // a failed copy: the staged arrays are rolled back
        if (rs != CL_OK) return rs;
    }
    tmr.lap("push: wait for the qualities");
    if (direct) c->q_dev += nq;
    undo.armed = false;
    return CL_OK;
}

cl_status cl_push_reads(cl_ctx *c, const cl_read_tile *t)
{
    // no exception leaves the library through the C ABI
    try { return cl_push_reads_impl(c, t); }
    catch (const std::bad_alloc &) { return fail(c, CL_ERR_NOMEM, "out of memory"); }
    catch (...) { return fail(c, CL_ERR_INVALID, "internal error"); }
}

static cl_status cl_push_reads_bits_impl(cl_ctx *c, const cl_read_tile_bits *b)
{
    if (!c || !b) return CL_ERR_INVALID;
    if (!c->in_contig || c->uploaded) return fail(c, CL_ERR_INVALID, "cl_push_reads_bits outside cl_contig_begin .. upload");
    if (!c->bits) return fail(c, CL_ERR_INVALID, "cl_push_reads_bits: this context runs the byte forms (DUT_QUAL_FORM=bytes), which need the quality bytes: use cl_push_reads");
    const uint64_t n = b->n_reads;
    if (n == 0) return CL_OK;
    if (!b->pos || !b->mapq || !b->cigar_off || !b->qual_off) return fail(c, CL_ERR_INVALID, "null tile array");
    if (c->h_pos.size() + n >= (1ull << 29)) return fail(c, CL_ERR_RANGE, "more than 2^29 reads in one contig");
    const uint32_t cig0 = b->cigar_off[0];
    const uint64_t q0 = b->qual_off[0];
    if (b->cigar_off[n] < cig0 || b->qual_off[n] < q0) return fail(c, CL_ERR_INVALID, "offset arrays must be non-decreasing");
    const uint64_t ncig = (uint64_t)b->cigar_off[n] - cig0, nq = b->qual_off[n] - q0;
    if (ncig && !b->cigar) return fail(c, CL_ERR_INVALID, "null cigar array");
    if (nq && (!b->pass_bits || !b->pass_sum)) return fail(c, CL_ERR_INVALID, "null pass_bits / pass_sum array");
    if (c->q_dev + nq >= (1ull << 38)) return fail(c, CL_ERR_RANGE, "more than 2^38 quality values in one contig");
    cl_read_tile t;
    t.n_reads = n; t.pos = b->pos; t.mapq = b->mapq; t.cigar_off = b->cigar_off; t.cigar = b->cigar; t.qual_off = b->qual_off; t.qual = nullptr;
    static const uint64_t kNoBits[2] = {0, 0};
    static const uint32_t kNoSum[1] = {0};
    return push_reads_bits(c, &t, cig0, q0, ncig, nq, b->pass_bits ? b->pass_bits : kNoBits, b->pass_sum ? b->pass_sum : kNoSum);
}

cl_status cl_push_reads_bits(cl_ctx *c, const cl_read_tile_bits *b)
{
    try { return cl_push_reads_bits_impl(c, b); }
    catch (const std::bad_alloc &) { return fail(c, CL_ERR_NOMEM, "out of memory"); }
    catch (...) { return fail(c, CL_ERR_INVALID, "internal error"); }
}


static cl_status cl_contig_upload_impl(cl_ctx *c)
{
    Range rg("cl_contig_upload");
    if (!c || !c->in_contig) return fail(c, CL_ERR_INVALID, "cl_contig_upload without cl_contig_begin");
    if (c->host_only) return fail(c, CL_ERR_DEVICE, "a host-only context (cl_debug_host_create) has no device to upload to");
    HIP_TRY(c, hipSetDevice(c->device));
    drop_prefetch(c);
    join_prealloc(c);
    c->n_reads = (uint32_t)c->h_pos.size();
    c->n_cigar = c->bits ? c->host_n_ops : c->h_cigar.size();
    if (!c->bits) {
        cl_status fs = flush_staged_qual(c);
        if (fs != CL_OK) return fs;
    }
    c->n_qual = c->q_dev;
    c->dev_sum_q = c->host_sum_q; c->dev_sum_cov = c->bits ? c->host_sum_cov : 0; c->dev_sum_mapq = c->bits ? c->host_sum_mapq : 0;
    c->n_wide = (uint32_t)c->h_wide_idx.size();
    const size_t n = c->n_reads;
    c->form = pick_form(c);
    const int form = c->form;
    StageTimer tmr0;
    // What the device needs of the per-read fields depends on the form of the pileup kernel the contig gets: the record
    // forms (pass bits: always; bytes: short reads) read records (built below) and nothing else per read; the run-table
    // form pos, mapq and end of the windows' candidates, and the table that the walk in size_for_extent() builds from the
    // staged CIGARs.  No form reads a CIGAR or an offset array: neither is uploaded.
    if (form == 2) HIP_TRY(c, c->d_end.reserve(n + 1));
    if (!c->bits) HIP_TRY(c, c->d_qual.grow_keep(c->n_qual + 2 * kQualPad, c->n_qual ? kQualPad + c->n_qual : 0, c->stream));
    HIP_TRY(c, c->d_wide_idx.reserve(c->n_wide + 1));
    tmr0.lap("upload: device buffers");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    tmr0.lap("upload: stream idle");
    StageTimer tmr;
    // everything goes through the pinned staging ring (pageable vectors -> pinned buffers -> HBM, the fills overlapping
    // the transfers)
    cl_status rs = CL_OK;
    if (form == 2) {
        HIP_TRY(c, c->d_pos.reserve(n + 1));
        HIP_TRY(c, c->d_mapq.reserve(n + 1));
        if ((rs = ring_copy(c, c->d_pos.p, c->h_pos.data(), n * sizeof(int32_t))) != CL_OK) return rs;
        if ((rs = ring_copy(c, c->d_mapq.p, c->h_mapq.data(), n)) != CL_OK) return rs;
        if ((rs = ring_copy(c, c->d_end.p, c->h_end.data(), n * sizeof(uint32_t))) != CL_OK) return rs;
    }
    c->h_rec_of.clear(); c->h_wide_rec_of.clear(); c->n_rec = 0;
    if (form == 3) {
        // the heads (kernels.hip.h): 8 bytes per read the pileup holds -- [pos, pos + span) and whether its mapq counts
        // as low (mod.rs:22-28) --, built straight into the pinned buffers; a span beyond kHeadSpanMax is cut into
        // several heads (h_rec_cnt, counted by cl_push_reads' walk)
        if ((rs = build_rec_index(c)) != CL_OK) return rs;
        tmr.lap("upload: record index");
        const uint32_t n_rec = c->n_rec;
        HIP_TRY(c, c->d_heads.reserve((size_t)n_rec + 1));
        const int32_t *hp = c->h_pos.data(); const uint8_t *hm = c->h_mapq.data(); const uint32_t *he = c->h_end.data();
        const uint32_t *ro = c->h_rec_of.data();
        const uint32_t max_low = c->opt.max_low_mapq, hs = c->head_span;
        rs = ring_start(c, reinterpret_cast<uint8_t *>(c->d_heads.p), ((uint64_t)n_rec + 1) * sizeof(uint2),
                        [hp, hm, he, ro, n, n_rec, max_low, hs](uint64_t off, uint64_t len, uint8_t *out) {
            uint2 *o = reinterpret_cast<uint2 *>(out);
            const uint64_t j0 = off / sizeof(uint2), j1 = (off + len) / sizeof(uint2);
            if (j1 > n_rec) memset(static_cast<void *>(o + (std::max<uint64_t>(n_rec, j0) - j0)), 0, (j1 - std::max<uint64_t>(n_rec, j0)) * sizeof(uint2));   // the padding head
            if (j0 >= n_rec) return;
            // the read that holds head j0: the last one whose range starts at or before it
            size_t i = (size_t)(std::upper_bound(ro, ro + n + 1, (uint32_t)j0) - ro) - 1;
            for (; i < n && ro[i] < j1; ++i) {
                const uint64_t jb = ro[i], je = ro[i + 1];
                const uint32_t low = (uint32_t)hm[i] <= max_low ? 0x80000000u : 0u;
                uint64_t x = (uint32_t)hp[i];
                const uint64_t e = he[i];
                for (uint64_t j = jb; j < je; ++j, x += hs) {
                    if (j < j0 || j >= j1) continue;
                    const uint64_t sp = std::min<uint64_t>(hs, e - x);
                    o[j - j0] = make_uint2((uint32_t)x, (uint32_t)sp | low);
                }
            }
        }, rec_chunk_bytes());
        if (rs == CL_OK) rs = ring_finish(c); else (void)ring_finish(c);
        if (rs != CL_OK) return rs;
        tmr.lap("upload: heads built + sent");
    } else if (form != 2) {
        // the records (kernels.hip.h: ReadRec): the host's walk over the CIGARs, so that the device decodes none --
        // north_star's "CIGAR-expanded ref spans" on the host side of the boundary.  Counted first (the reads' record
        // ranges are what the windows' candidate ranges index), then built straight into the pinned buffers: a buffer
        // covers a range of record numbers, the reads it belongs to are found by binary search.  Pass-bit form: a head
        // record per read and nothing else (its M/=/X runs are in the rows).
        if ((rs = build_rec_index(c)) != CL_OK) return rs;
        tmr.lap("upload: record index");
        const uint32_t n_rec = c->n_rec;
        HIP_TRY(c, c->d_rec.reserve((size_t)n_rec + 1));
        const int32_t *hp = c->h_pos.data(); const uint8_t *hm = c->h_mapq.data(); const uint32_t *he = c->h_end.data();
        const uint32_t *hc = c->h_cigar_off.data(), *hcig = c->h_cigar.data(); const unsigned long long *hq = c->h_qual_off.data();
        const uint32_t *ro = c->h_rec_of.data();
        const uint32_t min_mapq = c->opt.min_mapping_quality;
        // byte form: k_pileup loads 16-byte units around a record's run [qoff, qoff + len): within 15 bytes of its ends,
        // which the padding of the quality array covers as long as the run itself lies inside [0, n_qual] -- checked for
        // every record as it is built (CL_ERR_RANGE instead of a launch that would fault)
        std::atomic<bool> rec_oor{false};
        std::atomic<bool> *roor = &rec_oor;
        const unsigned long long nq_all = c->n_qual;
        const size_t inject_read = fault_injected("rec") ? n / 2 : (size_t)-1;
        rs = ring_start(c, reinterpret_cast<uint8_t *>(c->d_rec.p), ((uint64_t)n_rec + 1) * sizeof(ReadRec),
                        [hp, hm, he, hc, hcig, hq, ro, n, n_rec, min_mapq, roor, nq_all, inject_read](uint64_t off, uint64_t len, uint8_t *out) {
            ReadRec *o = reinterpret_cast<ReadRec *>(out);
            const uint64_t j0 = off / sizeof(ReadRec), j1 = (off + len) / sizeof(ReadRec);
            if (j1 > n_rec) memset(static_cast<void *>(o + (std::max<uint64_t>(n_rec, j0) - j0)), 0, (j1 - std::max<uint64_t>(n_rec, j0)) * sizeof(ReadRec));   // the padding record
            if (j0 >= n_rec) return;
            // the read that holds record j0: the last one whose range starts at or before it
            size_t i = (size_t)(std::upper_bound(ro, ro + n + 1, (uint32_t)j0) - ro) - 1;
            for (; i < n && ro[i] < j1; ++i) {
                const uint64_t jb = ro[i];
                const unsigned long long q0i = hq[i] + (i == inject_read ? 0x7FFFFFF0ull : 0ull);
                if (q0i + (hq[i + 1] - hq[i]) > nq_all) roor->store(true, std::memory_order_relaxed);   // (runs lie inside the read's bytes)
                gen_read_recs(hp[i], he[i], hm[i], min_mapq, hcig + hc[i], hc[i + 1] - hc[i], q0i, hq[i + 1] - hq[i],
                              [&](uint32_t k, const ReadRec &r) { const uint64_t j = jb + k; if (j >= j0 && j < j1) o[j - j0] = r; });
            }
        }, rec_chunk_bytes());
        if (rs == CL_OK) rs = ring_finish(c); else (void)ring_finish(c);
        if (rs != CL_OK) return rs;
        if (rec_oor.load()) return fail(c, CL_ERR_RANGE, "a read record addresses quality bytes outside the resident array");
        tmr.lap("upload: records built + sent");
    }
    std::vector<uint32_t> wide_rec;                      // record forms: the wide reads' records, read by read
    if (c->n_wide && form != 2) {
        c->h_wide_rec_of.assign(c->n_wide + 1, 0u);
        for (uint32_t j = 0; j < c->n_wide; ++j) {
            const uint32_t i = c->h_wide_idx[j];
            for (uint32_t r = c->h_rec_of[i]; r < c->h_rec_of[i + 1]; ++r) wide_rec.push_back(r);
            c->h_wide_rec_of[j + 1] = (uint32_t)wide_rec.size();
        }
        HIP_TRY(c, c->d_wide_idx.reserve(wide_rec.size() + 1));
        if (!wide_rec.empty())
            HIP_TRY(c, hipMemcpyAsync(c->d_wide_idx.p, wide_rec.data(), wide_rec.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    } else if (c->n_wide) {
        HIP_TRY(c, hipMemcpyAsync(c->d_wide_idx.p, c->h_wide_idx.data(), c->n_wide * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    }
    if (!c->bits && c->d_qual.p) {
        HIP_TRY(c, hipMemsetAsync(c->d_qual.p, 0, kQualPad, c->stream));
        HIP_TRY(c, hipMemsetAsync(c->d_qual.p + kQualPad + c->n_qual, 0, kQualPad, c->stream));
    }
    HIP_TRY(c, hipMemsetAsync(c->d_errflag.p, 0, 2 * sizeof(uint32_t), c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // a read overhanging the contig end makes the reference walk (and classify as REF_N, mod.rs:100-101) positions
    // up to its end: the extent is known from the spans computed at cl_push_reads
    tmr.lap("upload: small + sync");
    cl_status s = size_for_extent(c, (uint32_t)std::max<uint64_t>(c->contig_len, c->host_max_end));
    if (s != CL_OK) return s;
    tmr.lap("upload: extent, ref, bounds");
    if (c->d_iv.cap == 0) HIP_TRY(c, c->d_iv.reserve(1u << 20));
    // The staged copy is no longer needed; its memory goes to the process's staging pool for the next contig, this
    // context's or another's (giving back and re-faulting a few hundred megabytes per contig was a fifth of a contig's
    // host time).
    c->h_pos.clear(); c->h_mapq.clear(); c->h_cigar.clear(); c->h_cigar_off.clear(); c->h_qual_off.clear(); c->h_ref.clear();
    c->h_end.clear(); c->h_ck_x.clear(); c->h_ck_y.clear(); c->h_qbits.clear(); c->h_rb_off.clear(); c->h_sc_off.clear(); c->h_sc.clear();
    c->h_rec_of.clear(); c->h_wide_rec_of.clear();       // (capacity kept for the context's next contig)
    c->h_rec_cnt.clear();
    give_staging(c);
    std::vector<uint8_t>().swap(c->h_qual);
    tmr.lap("upload: done");
    c->uploaded = true; c->ran = false;
    return CL_OK;
}

cl_status cl_contig_upload(cl_ctx *c)
{
    // no exception leaves the library through the C ABI
    try { return cl_contig_upload_impl(c); }
    catch (const std::bad_alloc &) { return fail(c, CL_ERR_NOMEM, "out of memory"); }
    catch (...) { return fail(c, CL_ERR_INVALID, "internal error"); }
}


cl_status cl_debug_read_records(int32_t pos, const uint32_t *cigar, uint32_t n_ops, uint8_t mapq, uint8_t min_mapping_quality,
                                uint64_t qual_off, uint64_t qual_len, uint32_t *out, uint32_t cap,
                                uint32_t *n_records, uint32_t *phase)
{
    if ((n_ops && !cigar) || (cap && !out) || !n_records) return CL_ERR_INVALID;
    // the read's end as cl_push_reads' walk takes it: pos + bam_cigar2rlen, the read spanning nothing beyond the range
    unsigned long long l = 0;
    for (uint32_t q = 0; q < n_ops; ++q) if ((0x18Du >> (cigar[q] & 15u)) & 1u) l += cigar[q] >> 4;
    if (pos < 0) return CL_ERR_INVALID;
    const uint32_t end = l <= 0xFFFF0000ull - (uint64_t)pos ? (uint32_t)((uint64_t)pos + l) : (uint32_t)pos;
    uint32_t ph = 0;
    const uint32_t n = gen_read_recs(pos, end, mapq, min_mapping_quality, cigar, n_ops, qual_off, qual_len,
                                     [&](uint32_t k, const ReadRec &r) {
                                         if (k < cap) { out[4 * k] = (uint32_t)r.pos; out[4 * k + 1] = r.span; out[4 * k + 2] = r.qual_lo; out[4 * k + 3] = r.meta; }
                                     }, &ph);
    *n_records = n;
    if (phase) *phase = ph;
    return CL_OK;
}

cl_status cl_debug_qual_pack(const uint8_t *qual, uint64_t n, uint8_t min_base_quality, int level, uint64_t *words_out, uint64_t *sum_out)
{
    if (level >= 10 && level <= 12) {                            // the one-pass form cl_push_reads uses per read
        if ((n && !qual) || !words_out) return CL_ERR_INVALID;
        const uint64_t sm = dut::qual_pass_read(qual, n, min_base_quality, words_out, level - 10);
        if (sum_out) *sum_out = sm;
        return CL_OK;
    }
    if ((n && !qual) || level < 0 || level > 2) return CL_ERR_INVALID;
    if (words_out) {
        dut::qual_pass_words(qual, n >> 6, min_base_quality, words_out, level);
        if (n & 63ull) words_out[n >> 6] = dut::qual_pass_partial(qual + (n & ~63ull), (uint32_t)(n & 63ull), min_base_quality);
    }
    if (sum_out) *sum_out = dut::qual_pass_sum(qual, n, min_base_quality, level);
    return CL_OK;
}

cl_status cl_debug_ref_n_bits(const uint8_t *ref, uint64_t n_bases, uint64_t n_words, int level, uint64_t *words_out)
{
    if ((n_bases && !ref) || (n_words && !words_out) || level < 0 || level > 2) return CL_ERR_INVALID;
    dut::ref_n_words(ref, n_bases, n_words, words_out, level);
    return CL_OK;
}

cl_status cl_debug_host_create(const cl_options *opt, cl_ctx **out)
{
    if (!opt || !out) return CL_ERR_INVALID;
    *out = nullptr;
    cl_ctx *c = new (std::nothrow) cl_ctx();
    if (!c) return CL_ERR_NOMEM;
    c->device = -1; c->opt = *opt; c->host_only = true; c->bits = true;
    *out = c;
    return CL_OK;
}

static cl_status cl_debug_pass_rows_impl(cl_ctx *c, uint32_t *n_groups, uint32_t n_win_cap, uint32_t *rows, uint64_t cap_words,
                                         uint64_t *n_words, uint32_t *n_windows, uint64_t *summed_baseq)
{
    if (!c || !c->in_contig || !c->bits) return fail(c, CL_ERR_INVALID, "cl_debug_pass_rows: no staged contig in the pass-bit form");
    const uint32_t extent = (uint32_t)std::max<uint64_t>(c->contig_len, c->host_max_end);
    const uint32_t n_win = (uint32_t)(((uint64_t)extent + kT - 1) / kT);
    c->n_win = n_win;
    if (n_windows) *n_windows = n_win;
    if (summed_baseq) *summed_baseq = c->host_sum_q;
    std::vector<WinMeta> win;
    uint32_t flags = 0;
    host_window_bounds(c, win, flags);
    const dut::RowReads H = row_reads(c);
    std::vector<dut::RowCur> act, save;
    dut::RowScratch sc;
    std::vector<uint32_t> buf;
    uint64_t used = 0;
    for (uint32_t w = 0; w < n_win; ++w) {
        const uint32_t W = w * kT;
        const WinMeta &m = win[w];
        // (every third window entered afresh, as the first window of a thread's range is -- through the checkpoints of the
        // long reads --, the others carried over from the window before, as inside a range)
        act.clear();
        if (w % 3u == 0u) {
            for (uint32_t i = 0; i < m.wn; ++i) dut::rows_enter(act, H, c->h_wide_idx[m.wlo + i], W);
            for (uint32_t r = m.lo; r < m.hi; ++r) dut::rows_enter(act, H, r, W);
        } else {
            act = save;
            for (uint32_t r = win[w - 1].hi; r < m.hi; ++r) dut::rows_enter(act, H, r, W);
        }
        size_t cap = 16, cnt;
        std::vector<dut::RowCur> start = act;
        for (;;) {
            buf.assign(cap * dut::kRowGroupWords, 0xDEADBEEFu);         // groups must be zeroed by the builder itself
            act = start;
            cnt = dut::rows_window<kT>(act, H, W, buf.data(), cap, sc);
            if (cnt != SIZE_MAX) break;
            cap *= 4;
        }
        save = act;
        if (w < n_win_cap && n_groups) n_groups[w] = (uint32_t)cnt;
        const uint64_t nw = (uint64_t)cnt * dut::kRowGroupWords;
        if (rows && used + nw <= cap_words) memcpy(rows + used, buf.data(), nw * sizeof(uint32_t));
        used += nw;
    }
    if (n_words) *n_words = used;
    return CL_OK;
}

cl_status cl_debug_pass_rows(cl_ctx *c, uint32_t *n_groups, uint32_t n_win_cap, uint32_t *rows, uint64_t cap_words,
                             uint64_t *n_words, uint32_t *n_windows, uint64_t *summed_baseq)
{
    try { return cl_debug_pass_rows_impl(c, n_groups, n_win_cap, rows, cap_words, n_words, n_windows, summed_baseq); }
    catch (const std::bad_alloc &) { return fail(c, CL_ERR_NOMEM, "out of memory"); }
    catch (...) { return fail(c, CL_ERR_INVALID, "internal error"); }
}

cl_status cl_contig_run(cl_ctx *c)
{
    Range rg("cl_contig_run");
    if (!c || !c->uploaded) return fail(c, CL_ERR_INVALID, "cl_contig_run before cl_contig_upload");
    HIP_TRY(c, hipSetDevice(c->device));
    cl_status s = enqueue(c, false, nullptr, nullptr, nullptr);
    if (s == CL_OK) c->ran = true;
    return s;
}

cl_status cl_sync(cl_ctx *c)
{
    if (!c) return CL_ERR_INVALID;
    if (c->host_only) return fail(c, CL_ERR_DEVICE, "a host-only context has no device");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return harvest_events(c);
}

static cl_status check_summary(cl_ctx *c)
{
    if ((c->h_sum.err | c->bounds_err | c->host_err) & kErrRange) return fail(c, CL_ERR_RANGE, "a read ends beyond the engine's 32-bit coordinate range");
    if ((c->h_sum.err | c->host_err) & kErrCigar)
        return fail(c, CL_ERR_CIGAR, "malformed CIGAR: zero-length reference-consuming operation, or a single non-match "
                                     "operation on a read that spans reference positions (undefined in htslib's pileup)");
    return CL_OK;
}

static cl_status cl_contig_collect_impl(cl_ctx *c, cl_contig_summary *out, const cl_interval **intervals, size_t *n_intervals)
{
    Range rg("cl_contig_collect");
    if (!c || !c->ran) return fail(c, CL_ERR_INVALID, "cl_contig_collect before cl_contig_run");
    HIP_TRY(c, hipSetDevice(c->device));
    StageTimer tmr;
    // every `continue` below re-runs the contig for one distinct reason (32-bit counters: once; 16-bit fields in the
    // marked windows: the marks are sticky, at most twice; a larger extent: once per overhang level), so a handful of
    // rounds always suffices -- if they do not, the device state and h_sum disagree and nothing may be returned
    bool converged = false;
    for (int attempt = 0; attempt < 8 && !converged; ++attempt) {
        HIP_TRY(c, hipMemcpyAsync(&c->h_sum, c->d_summary.p, sizeof(DevSummary), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        cl_status s = harvest_events(c);
        if (s != CL_OK) return s;
        s = check_summary(c);
        if (s != CL_OK) return s;
        // a window touched by more reads than the 16-bit counters hold: redo with 32-bit counters
        if ((c->h_sum.err & kNeedDeep) && !c->deep) {
            c->deep = true;
            s = enqueue(c, false, nullptr, nullptr, nullptr);
            if (s != CL_OK) return s;
            continue;
        }
        // a position deeper than 255 in a window that used the 8-bit counter sets beyond their safe
        // candidate count: the kernel marked those windows, run again (they now use 16-bit fields)
        if ((c->h_sum.err & kNeedWide8) && !c->deep) {             // the marks are sticky: at most one more run raises it
            s = enqueue(c, false, nullptr, nullptr, nullptr);
            if (s != CL_OK) return s;
            continue;
        }
        // (a read that overhangs the contig end makes the reference walk, and classify as REF_N, positions up to its
        // end, mod.rs:100-101: the extent was sized for that at upload from the ends computed at cl_push_reads)
        if (c->h_sum.n_intervals > c->d_iv.cap) {
            HIP_TRY(c, c->d_iv.reserve(c->h_sum.n_intervals));
            launch_tail(c);
            HIP_TRY(c, hipGetLastError());
        }
        converged = true;
    }
    if (!converged) return fail(c, CL_ERR_DEVICE, "cl_contig_collect: the re-run loop (counter width / extent) did not converge");
    tmr.lap("collect: kernels + summary");
    const size_t niv = c->h_sum.n_intervals;
    static_assert(sizeof(cl_interval) == sizeof(Interval), "interval layout");
    try { c->h_iv.resize(niv); } catch (const std::bad_alloc &) { return fail(c, CL_ERR_NOMEM, "interval buffer"); }
    if (niv) {
        HIP_TRY(c, hipMemcpyAsync(c->h_iv.data(), c->d_iv.p, niv * sizeof(Interval), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    tmr.lap("collect: intervals D2H");
    if (out) {
        static_assert(offsetof(DevSummary, max_end) == sizeof(cl_contig_summary), "summary layout");
        memcpy(out, &c->h_sum, sizeof(cl_contig_summary));
    }
    if (intervals) *intervals = c->h_iv.data();
    if (n_intervals) *n_intervals = niv;
    return CL_OK;
}

cl_status cl_contig_collect(cl_ctx *c, cl_contig_summary *out, const cl_interval **intervals, size_t *n_intervals)
{
    // no exception leaves the library through the C ABI
    try { return cl_contig_collect_impl(c, out, intervals, n_intervals); }
    catch (const std::bad_alloc &) { return fail(c, CL_ERR_NOMEM, "out of memory"); }
    catch (...) { return fail(c, CL_ERR_INVALID, "internal error"); }
}


cl_status cl_contig_finish(cl_ctx *c, cl_contig_summary *out, const cl_interval **intervals, size_t *n_intervals)
{
    StageTimer tmr;
    cl_status s = cl_contig_upload(c);
    if (s != CL_OK) return s;
    tmr.lap("finish: upload");
    s = cl_contig_run(c);
    if (s != CL_OK) return s;
    tmr.lap("finish: run (launches)");
    s = cl_contig_collect(c, out, intervals, n_intervals);
    tmr.lap("finish: collect");
    return s;
}

cl_status cl_contig_abort(cl_ctx *c)
{
    if (!c) return CL_ERR_INVALID;
    if (!c->host_only) {
        (void)hipSetDevice(c->device);
        drop_prefetch(c);                                 // joins the copiers: nothing reads the caller's buffer any more
        join_prealloc(c);
        if (c->stream) (void)hipStreamSynchronize(c->stream);
    }
    c->h_pos.clear(); c->h_mapq.clear(); c->h_cigar.clear(); c->h_cigar_off.clear(); c->h_qual_off.clear(); c->h_ref.clear();
    c->h_end.clear(); c->h_ck_x.clear(); c->h_ck_y.clear(); c->h_qual.clear(); c->h_qbits.clear(); c->h_rec_cnt.clear();
    c->h_rb_off.clear(); c->h_sc_off.clear(); c->h_sc.clear();
    c->q_dev = 0; c->host_sum_q = 0; c->host_n_ops = 0; c->host_sum_cov = 0; c->host_sum_mapq = 0;
    c->in_contig = false; c->uploaded = false; c->ran = false;
    return CL_OK;
}

cl_status cl_device_summary(cl_ctx *c, void **dev_ptr, size_t *bytes)
{
    if (!c || !dev_ptr || !bytes) return CL_ERR_INVALID;
    *dev_ptr = c->d_summary.p;
    *bytes = sizeof(cl_contig_summary);
    return CL_OK;
}

cl_status cl_set_profiling(cl_ctx *c, int on)
{
    if (!c) return CL_ERR_INVALID;
    c->profiling = on != 0;
    return CL_OK;
}

cl_status cl_get_kernel_ms(cl_ctx *c, double ms[CL_K_COUNT], uint64_t *n_runs)
{
    if (!c) return CL_ERR_INVALID;
    cl_status s = harvest_events(c);
    if (s != CL_OK) return s;
    if (ms) for (int i = 0; i < CL_K_COUNT; ++i) ms[i] = c->ms[i];
    if (n_runs) *n_runs = c->n_runs;
    return CL_OK;
}

cl_status cl_reset_kernel_ms(cl_ctx *c)
{
    if (!c) return CL_ERR_INVALID;
    cl_status s = harvest_events(c);
    if (s != CL_OK) return s;
    for (int i = 0; i < CL_K_COUNT; ++i) c->ms[i] = 0.0;
    c->n_runs = 0;
    return CL_OK;
}

cl_status cl_contig_bytes(cl_ctx *c, uint64_t *input_bytes, uint64_t *output_bytes)
{
    if (!c || !c->uploaded) return fail(c, CL_ERR_INVALID, "no resident contig");
    // What one run of the resident form must read at least once, counted strictly: the array elements the form's kernel
    // addresses, nothing it does not (no CIGAR word: none is resident in any form), and what it must write: the intervals
    // (12 bytes each; the per-position counters and states never reach HBM).
    //   every form   reference bytes (extent; pass bits: one bit per position) + one 32-byte window record per window
    //   pass bits    the rows (1 KB per group of 4 rows) + an 8-byte head per read with a span + the wide list
    //   bytes, 0     the quality bytes + the 16-byte records (heads and pieces) + the wide list
    //   bytes, 2     the quality bytes + 8 bytes per piece of the run table + pos 4, end 4, mapq 1 per read + the wide list
    uint64_t in = (c->form == 3 ? ((uint64_t)c->extent + 7) / 8 : (uint64_t)c->extent) + (uint64_t)c->n_win * sizeof(WinMeta);
    if (c->form == 3) in += c->n_row_groups * (uint64_t)(dut::kRowGroupWords * sizeof(uint32_t)) + (uint64_t)c->n_rec * sizeof(uint2) + (uint64_t)c->n_wide * 4;
    else if (c->form == 0) in += c->n_qual + (uint64_t)c->n_rec * sizeof(ReadRec) + (uint64_t)c->n_wide * 4;
    else in += c->n_qual + c->n_runtab * 8 + (uint64_t)c->n_reads * 9 + (uint64_t)c->n_wide * 4;
    if (input_bytes) *input_bytes = in;
    if (output_bytes) *output_bytes = 12ull * c->h_sum.n_intervals;
    return CL_OK;
}

cl_status cl_contig_layout(cl_ctx *c, cl_layout_info *out)
{
    if (!c || !out || !c->uploaded) return fail(c, CL_ERR_INVALID, "no resident contig");
    memset(out, 0, sizeof(*out));
    out->form = c->form;
    out->n_reads = c->n_reads; out->n_records = c->n_rec; out->n_windows = c->n_win;
    out->n_qual = c->n_qual; out->n_cigar = c->n_cigar;
    out->row_groups = c->n_row_groups; out->max_groups = c->max_groups;
    out->run_table_entries = c->n_runtab;
    out->counter_planes = c->form == 3 ? (c->max_groups <= 63u ? 8u : c->max_groups <= 16383u ? 16u : 32u) : 0u;
    // HBM this context holds (the capacity of every device buffer: what cl_destroy gives back)
    uint64_t b = 0;
    b += c->d_pos.cap * 4 + c->d_mapq.cap + c->d_qual.cap + c->d_ref.cap + c->d_end.cap * 4 + c->d_rec.cap * sizeof(ReadRec) + c->d_heads.cap * sizeof(uint2) + c->d_refn.cap * 4;
    b += c->d_rows.cap * sizeof(uint4) + c->d_win_off.cap * 4 + c->d_wide_idx.cap * 4 + c->d_win.cap * sizeof(WinMeta);
    b += c->d_state.cap + c->d_runs.cap * 2 + c->d_first_state.cap + c->d_last_state.cap + c->d_win_wide.cap;
    b += c->d_winpart.cap * sizeof(WinPartial) + c->d_fin.cap * sizeof(FinPartial) + c->d_errflag.cap * 4 + c->d_runtab.cap * 8;
    b += c->d_lut.cap * 4 + c->d_summary.cap * sizeof(DevSummary) + c->d_iv.cap * sizeof(Interval) + c->d_dbg.cap * 4;
    out->device_bytes = b;
    // what cl_contig_upload sent over the link for this contig (every transfer goes through the pinned staging ring)
    const uint64_t padded = (uint64_t)c->n_win * kT + 16;
    uint64_t h = (c->form == 3 ? ((uint64_t)c->n_win * kT) / 8 : padded) + (uint64_t)c->n_win * sizeof(WinMeta) + (uint64_t)c->n_wide * 4;
    if (c->form == 3) h += c->n_row_groups * (uint64_t)(dut::kRowGroupWords * sizeof(uint32_t)) + ((uint64_t)c->n_rec + 1) * sizeof(uint2);
    else if (c->form == 0) h += c->n_qual + ((uint64_t)c->n_rec + 1) * sizeof(ReadRec);
    else h += c->n_qual + c->n_runtab * 8 + (uint64_t)c->n_reads * 9;
    out->upload_h2d_bytes = h;
    return CL_OK;
}

static cl_status cl_debug_depths_impl(cl_ctx *c, uint32_t *raw, uint32_t *qc, uint32_t *low, uint8_t *state, uint64_t cap)
{
    if (!c || !c->ran) return fail(c, CL_ERR_INVALID, "cl_debug_depths needs a collected contig");
    HIP_TRY(c, hipSetDevice(c->device));
    if (cap < c->extent) return fail(c, CL_ERR_INVALID, "cap < extent");
    const size_t padded = (size_t)c->n_win * kT;
    HIP_TRY(c, c->d_dbg.reserve(3 * padded + 1));
    cl_status s = enqueue(c, true, c->d_dbg.p, c->d_dbg.p + padded, c->d_dbg.p + 2 * padded);
    if (s != CL_OK) return s;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const size_t nb = (size_t)c->extent * sizeof(uint32_t);
    if (raw && nb) HIP_TRY(c, hipMemcpy(raw, c->d_dbg.p, nb, hipMemcpyDeviceToHost));
    if (qc && nb) HIP_TRY(c, hipMemcpy(qc, c->d_dbg.p + padded, nb, hipMemcpyDeviceToHost));
    if (low && nb) HIP_TRY(c, hipMemcpy(low, c->d_dbg.p + 2 * padded, nb, hipMemcpyDeviceToHost));
    if (state && c->extent) HIP_TRY(c, hipMemcpy(state, c->d_state.p, c->extent, hipMemcpyDeviceToHost));
    return CL_OK;
}

cl_status cl_debug_depths(cl_ctx *c, uint32_t *raw, uint32_t *qc, uint32_t *low, uint8_t *state, uint64_t cap)
{
    // no exception leaves the library through the C ABI
    try { return cl_debug_depths_impl(c, raw, qc, low, state, cap); }
    catch (const std::bad_alloc &) { return fail(c, CL_ERR_NOMEM, "out of memory"); }
    catch (...) { return fail(c, CL_ERR_INVALID, "internal error"); }
}


// the site list as the kernel wants it: sorted by 0-based position (vcf_pos - 1, caller.rs:94) with the original
// indices, vcf_pos 0 left out (it can never match), and the first sorted site at or after every 256th position
struct SitePrep { std::vector<uint32_t> pos0, idx, bucket; uint32_t n_buckets = 0; };
static void site_prepare(const uint32_t *sites, size_t n_sites, SitePrep &P)
{
    // (position, original index) sorted by position, ties by index: a stable radix sort on the 32-bit positions, three
    // passes of 11 bits (std::sort took 10 of the 11 ms of a run on a resident tile with 200 000 sites)
    std::vector<unsigned long long> key(n_sites), tmp(n_sites);
    for (size_t i = 0; i < n_sites; ++i) key[i] = ((unsigned long long)sites[i] << 32) | (unsigned long long)i;
    for (int pass = 0; pass < 3; ++pass) {
        const int sh = 32 + 11 * pass;
        size_t cnt[2049] = {0};
        for (size_t i = 0; i < n_sites; ++i) cnt[((key[i] >> sh) & 2047u) + 1] += 1;
        for (int b = 0; b < 2048; ++b) cnt[b + 1] += cnt[b];
        for (size_t i = 0; i < n_sites; ++i) tmp[cnt[(key[i] >> sh) & 2047u]++] = key[i];
        key.swap(tmp);
    }
    P.pos0.reserve(n_sites); P.idx.reserve(n_sites);
    for (size_t i = 0; i < n_sites; ++i) {
        const uint32_t s = (uint32_t)(key[i] >> 32);
        if (s == 0) continue;
        P.pos0.push_back(s - 1); P.idx.push_back((uint32_t)key[i]);
    }
    if (P.pos0.empty()) return;
    P.n_buckets = (uint32_t)(((uint64_t)P.pos0.back() >> 8) + 2);
    P.bucket.resize(P.n_buckets);
    size_t j = 0;
    for (uint32_t bk = 0; bk < P.n_buckets; ++bk) {
        while (j < P.pos0.size() && P.pos0[j] < ((uint64_t)bk << 8)) ++j;
        P.bucket[bk] = (uint32_t)j;
    }
}


// cl_site_pileup knows the site list when the tile is uploaded: only the reads that can add to the histogram travel --
// those the kernel itself would walk (k_site_pileup: position inside the contig, mapq >= min_quality, a site inside
// [pos, pos + reference span)); at one site per ~300 bases that is two short reads in five, and the bases are what the
// call spends its time sending (1.1 GB at the link's rate for BASELINE configs[4]).  The kept reads' records, CIGAR words
// and base BYTES are gathered straight into the pinned buffers; a read keeps its nibble parity (its bytes are copied
// whole), so read lengths can no longer be taken from offset differences: tiles with a read of 65 535 bases or 255
// operations and more (the records' escape values) are sent whole instead.
struct SiteGather {
    dut::Scratch<uint32_t> kidx;               // kept read k = read kidx[k] of the tile
    dut::Scratch<unsigned long long> B;        // K + 1: first byte of kept read k in the gathered base array
    dut::Scratch<uint32_t> coff;               // K + 1: first CIGAR word of kept read k in the gathered CIGAR array
    uint64_t K = 0;
    bool on = false;
};
static void site_filter(const cl_site_tile *t, const SitePrep &P, uint8_t min_quality, uint32_t contig_len, SiteGather &G)
{
    const uint64_t n = t->n_reads;
    G.on = false;
    if (n == 0 || P.pos0.empty()) return;
    const size_t grain = 1u << 16, nchunk = (n + grain - 1) / grain;
    dut::Scratch<uint8_t> keep(n);
    std::vector<uint64_t> c_k(nchunk + 1, 0), c_b(nchunk + 1, 0), c_c(nchunk + 1, 0);
    std::atomic<bool> escape{false};
    const uint32_t *pos0 = P.pos0.data(); const uint32_t *bucket = P.bucket.data();
    const uint32_t n_sites = (uint32_t)P.pos0.size(), n_buckets = P.n_buckets;
    uint8_t *kp = keep.get();
    dut::parallel_for(nchunk, 1, [&](size_t ch) {
        const size_t a = ch * grain, b = std::min<size_t>(n, a + grain);
        uint64_t k = 0, nb = 0, nc = 0;
        for (size_t i = a; i < b; ++i) {
            kp[i] = 0;
            const uint32_t c0 = t->cigar_off[i], c1 = t->cigar_off[i + 1];
            const uint64_t s0 = t->seq_off[i], s1 = t->seq_off[i + 1];
            if (c1 < c0 || s1 < s0) { escape.store(true); continue; }          // (refused by the upload's own check)
            if (c1 - c0 >= 255u || s1 - s0 >= 0xFFFFull) escape.store(true);
            if ((uint32_t)t->pos[i] >= contig_len || t->mapq[i] < min_quality) continue;
            unsigned long long reflen = 0;
            for (uint32_t q = c0; q < c1; ++q) { const uint32_t cw = t->cigar[q]; reflen += ((0x18Du >> (cw & 15u)) & 1u) ? (cw >> 4) : 0u; }
            const unsigned long long x = (uint32_t)t->pos[i];
            const unsigned long long bx = x >> 8;
            uint32_t lo = bx < n_buckets ? bucket[bx] : n_sites;
            while (lo < n_sites && pos0[lo] < x) ++lo;
            if (lo < n_sites && pos0[lo] < x + reflen) { kp[i] = 1; ++k; nb += ((s1 + 1) >> 1) - (s0 >> 1); nc += c1 - c0; }
        }
        c_k[ch + 1] = k; c_b[ch + 1] = nb; c_c[ch + 1] = nc;
    });
    if (escape.load()) return;
    for (size_t ch = 0; ch < nchunk; ++ch) { c_k[ch + 1] += c_k[ch]; c_b[ch + 1] += c_b[ch]; c_c[ch + 1] += c_c[ch]; }
    const uint64_t K = c_k[nchunk];
    if (c_c[nchunk] > 0xFFFFFFF0ull) return;
    G.kidx = dut::Scratch<uint32_t>(K + 1); G.B = dut::Scratch<unsigned long long>(K + 1); G.coff = dut::Scratch<uint32_t>(K + 1);
    uint32_t *kidx = G.kidx.get(); unsigned long long *B = G.B.get(); uint32_t *coff = G.coff.get();
    dut::parallel_for(nchunk, 1, [&](size_t ch) {
        const size_t a = ch * grain, b = std::min<size_t>(n, a + grain);
        uint64_t k = c_k[ch], nb = c_b[ch], nc = c_c[ch];
        for (size_t i = a; i < b; ++i) {
            if (!kp[i]) continue;
            kidx[k] = (uint32_t)i; B[k] = nb; coff[k] = (uint32_t)nc;
            nb += ((t->seq_off[i + 1] + 1) >> 1) - (t->seq_off[i] >> 1); nc += t->cigar_off[i + 1] - t->cigar_off[i];
            ++k;
        }
    });
    kidx[K] = 0; B[K] = c_b[nchunk]; coff[K] = (uint32_t)c_c[nchunk];
    G.K = K; G.on = true;
}

// ---- config 5: the tile goes to HBM once (cl_site_upload: packed records built straight into the pinned buffers, the
//      4-bit bases and the CIGAR words through the staging ring) and stays resident; any number of site lists can then be
//      run over it (cl_site_run).  cl_site_pileup is the two in one call. ----
static cl_status cl_site_upload_impl(cl_ctx *c, uint32_t contig_len, uint64_t ref_len, const cl_site_tile *t, const SiteGather *G = nullptr)
{
    if (!c || !t) return CL_ERR_INVALID;
    if (c->host_only) return fail(c, CL_ERR_DEVICE, "a host-only context has no device");
    HIP_TRY(c, hipSetDevice(c->device));
    drop_prefetch(c);                                        // the ring is needed below
    SiteResident &S = c->site;
    S.resident = false; S.filtered = false;
    const uint64_t n_all = t->n_reads;
    if (n_all > 0xFFFFFFF0ull) return fail(c, CL_ERR_RANGE, "too many reads");
    if (n_all && (!t->pos || !t->mapq || !t->cigar_off || !t->seq_off)) return fail(c, CL_ERR_INVALID, "null tile array");
    StageTimer tmr;
    // what travels: the whole tile, or (cl_site_pileup, site_filter above) the reads that overlap a site of its list
    const bool g = G && G->on;
    if (!g) {                                                // (site_filter has looked at every offset pair already)
        std::atomic<int> bad{0};
        dut::parallel_for(n_all, 262144, [&](size_t i) { if (t->cigar_off[i + 1] < t->cigar_off[i] || t->seq_off[i + 1] < t->seq_off[i]) bad = 1; });
        if (bad) return fail(c, CL_ERR_INVALID, "offset arrays must be non-decreasing");
    }
    const uint64_t n = g ? G->K : n_all;
    const uint32_t *kidx = g ? G->kidx.get() : nullptr;
    const unsigned long long *GB = g ? G->B.get() : nullptr;
    const uint32_t *gco = g ? G->coff.get() : nullptr;
    const uint64_t ncig = g ? gco[n] : (n_all ? t->cigar_off[n_all] : 0);
    const uint64_t nbytes = g ? GB[n] : ((n_all ? t->seq_off[n_all] : 0) + 1) / 2;
    const uint64_t nbase = g ? 2 * GB[n] : (n_all ? t->seq_off[n_all] : 0);
    const uint64_t *hs_all = t->seq_off;
    // base offset of (kept) read k in the array that travels; k = n: its end
    auto seq_at = [=](uint64_t k) -> uint64_t { return !g ? hs_all[k] : (k < n ? 2 * GB[k] + (hs_all[kidx[k]] & 1ull) : nbase); };
    const uint64_t n_blocks = (n + kBlock - 1) / kBlock;
    // a workgroup's reads must lie within 2^32 bases of its first one (256 reads: always, short of 16 M-base reads)
    for (uint64_t b = 0; b < n_blocks; ++b)
        if (seq_at(std::min<uint64_t>(n, (b + 1) * kBlock)) - seq_at(b * kBlock) > 0xFFFF0000ull)
            return fail(c, CL_ERR_RANGE, "reads too long for the site pileup");
    HIP_TRY(c, S.rec.reserve(n + 1)); HIP_TRY(c, S.base.reserve(n_blocks + 1));
    HIP_TRY(c, S.cig.reserve(ncig + 8)); HIP_TRY(c, S.seq.reserve(nbytes + 16));
    tmr.lap("site upload: checks + device buffers");
    cl_status rs = CL_OK;
    // the bases: the bulk of the tile (0.5 byte per aligned base)
    if (nbytes && !g && (rs = ring_copy(c, S.seq.p, t->seq4, nbytes)) != CL_OK) return rs;
    if (nbytes && g) {
        const uint8_t *seq4 = t->seq4;
        rs = ring_start(c, S.seq.p, nbytes, [seq4, hs_all, kidx, GB, n](uint64_t off, uint64_t len, uint8_t *out) {
            // the kept reads whose bytes fall into [off, off + len): whole bytes of the tile's array, read by read
            uint64_t k = (uint64_t)(std::upper_bound(GB, GB + n + 1, (unsigned long long)off) - GB) - 1;
            uint64_t at = off;
            const uint64_t end = off + len;
            while (at < end && k < n) {
                const uint64_t src0 = hs_all[kidx[k]] >> 1, take = std::min<uint64_t>(GB[k + 1], end) - at;
                memcpy(out + (at - off), seq4 + src0 + (at - GB[k]), take);
                at += take;
                if (at == GB[k + 1]) ++k;
            }
        }, PinRing::kPinBytes, PinRing::kCopyThreads);
        if (rs == CL_OK) rs = ring_finish(c); else (void)ring_finish(c);
        if (rs != CL_OK) return rs;
    }
    tmr.lap("site upload: bases");
    // one packed record per read (+ the sentinel with the totals), built in the pinned buffers
    {
        const int32_t *hp = t->pos; const uint8_t *hm = t->mapq; const uint32_t *hc = t->cigar_off;
        const uint64_t ncig_all = ncig;
        rs = ring_start(c, reinterpret_cast<uint8_t *>(S.rec.p), (n + 1) * sizeof(SiteRec), [=](uint64_t off, uint64_t len, uint8_t *out) {
            SiteRec *o = reinterpret_cast<SiteRec *>(out);
            const size_t i0 = off / sizeof(SiteRec), i1 = (off + len) / sizeof(SiteRec);
            for (size_t k = i0; k < i1; ++k) {
                SiteRec r;
                if (k < n) {
                    const size_t i = g ? kidx[k] : k;
                    const uint32_t nc = hc[i + 1] - hc[i];
                    const uint64_t sl = hs_all[i + 1] - hs_all[i];
                    r.pos = hp[i]; r.cigar_off = g ? gco[k] : hc[i]; r.seq_lo = (uint32_t)seq_at(k);
                    r.meta = (uint32_t)hm[i] | (std::min<uint32_t>(nc, 255u) << 8) | ((uint32_t)std::min<uint64_t>(sl, 0xFFFFull) << 16);
                } else { r.pos = 0; r.cigar_off = (uint32_t)ncig_all; r.seq_lo = (uint32_t)nbase; r.meta = 0; }
                o[k - i0] = r;
            }
        }, PinRing::kPinBytes, g ? PinRing::kCopyThreads : 0);
        // ... beside it, the 64-bit base offset of every workgroup's first read
        std::vector<unsigned long long> h_base(n_blocks + 1);
        for (uint64_t b = 0; b < n_blocks; ++b) h_base[b] = seq_at(b * kBlock);
        h_base[n_blocks] = nbase;
        if (rs == CL_OK) rs = ring_finish(c); else (void)ring_finish(c);
        if (rs != CL_OK) return rs;
        HIP_TRY(c, hipMemcpyAsync(S.base.p, h_base.data(), (n_blocks + 1) * sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    if (ncig && !g && (rs = ring_copy(c, S.cig.p, t->cigar, ncig * 4)) != CL_OK) return rs;
    if (ncig && g) {
        const uint32_t *cig = t->cigar; const uint32_t *hc = t->cigar_off;
        rs = ring_start(c, reinterpret_cast<uint8_t *>(S.cig.p), ncig * 4, [cig, hc, kidx, gco, n](uint64_t off, uint64_t len, uint8_t *out) {
            const uint64_t w0 = off / 4, w1 = (off + len) / 4;                  // (a buffer is a whole number of words)
            uint64_t k = (uint64_t)(std::upper_bound(gco, gco + n + 1, (uint32_t)w0) - gco) - 1;
            uint64_t at = w0;
            uint32_t *o = reinterpret_cast<uint32_t *>(out);
            while (at < w1 && k < n) {
                const uint64_t take = std::min<uint64_t>(gco[k + 1], w1) - at;
                const uint32_t *src = cig + hc[kidx[k]] + (at - gco[k]);
                uint32_t *dstw = o + (at - w0);
                for (uint64_t q = 0; q < take; ++q) dstw[q] = src[q];          // (a read has a word or three)
                at += take;
                if (at == gco[k + 1]) ++k;
            }
        }, PinRing::kPinBytes, PinRing::kCopyThreads);
        if (rs == CL_OK) rs = ring_finish(c); else (void)ring_finish(c);
        if (rs != CL_OK) return rs;
    }
    tmr.lap("site upload: records + cigar");
    S.n = n; S.ncig = ncig; S.nbase = nbase; S.contig_len = contig_len; S.ref_len = ref_len;
    S.resident = true; S.filtered = g;
    return CL_OK;
}

static cl_status cl_site_run_impl(cl_ctx *c, uint8_t min_quality, const uint32_t *sites, size_t n_sites, uint32_t *hist, const SitePrep *ready)
{
    if (!c || (!sites && n_sites) || (!hist && n_sites)) return CL_ERR_INVALID;
    if (c->host_only) return fail(c, CL_ERR_DEVICE, "a host-only context has no device");
    SiteResident &S = c->site;
    if (!S.resident) return fail(c, CL_ERR_INVALID, "cl_site_run without cl_site_upload");
    // (a tile that cl_site_pileup filtered for its own list serves that call only: ready != nullptr is that call)
    if (S.filtered && !ready) return fail(c, CL_ERR_INVALID, "cl_site_run: the resident tile was uploaded by cl_site_pileup for its own site list; cl_site_upload gives a tile that serves any list");
    HIP_TRY(c, hipSetDevice(c->device));
    if (n_sites == 0) return CL_OK;
    if (n_sites > 0x0FFFFFFFu) return fail(c, CL_ERR_RANGE, "too many sites");
    StageTimer tmr;
    SitePrep mine;
    if (!ready) { site_prepare(sites, n_sites, mine); ready = &mine; }
    const std::vector<uint32_t> &pos0 = ready->pos0, &idx = ready->idx, &bucket = ready->bucket;
    const uint32_t n_buckets = ready->n_buckets;
    memset(hist, 0, n_sites * 16 * sizeof(uint32_t));
    if (S.n == 0 || pos0.empty()) return CL_OK;
    tmr.lap("site run: sort + buckets");
    HIP_TRY(c, S.p0.reserve(pos0.size())); HIP_TRY(c, S.ix.reserve(pos0.size())); HIP_TRY(c, S.hist.reserve(n_sites * 16));
    HIP_TRY(c, S.bk.reserve(n_buckets));
    HIP_TRY(c, hipMemsetAsync(S.hist.p, 0, n_sites * 16 * 4, c->stream));
    HIP_TRY(c, hipMemcpyAsync(S.p0.p, pos0.data(), pos0.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(S.ix.p, idx.data(), idx.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(S.bk.p, bucket.data(), (size_t)n_buckets * 4, hipMemcpyHostToDevice, c->stream));
    SiteArgs A;
    A.rec = S.rec.p; A.seq_base = S.base.p; A.cigar = S.cig.p; A.seq4 = S.seq.p; A.n = (uint32_t)S.n;
    A.min_quality = min_quality; A.contig_len = S.contig_len; A.ref_len = S.ref_len;
    A.sorted_pos0 = S.p0.p; A.sorted_idx = S.ix.p; A.bucket = S.bk.p; A.n_buckets = n_buckets; A.n_sites = (uint32_t)pos0.size();
    A.hist = S.hist.p;
    if (!c->site_ev[0]) { HIP_TRY(c, hipEventCreate(&c->site_ev[0])); HIP_TRY(c, hipEventCreate(&c->site_ev[1])); }
    HIP_TRY(c, hipEventRecord(c->site_ev[0], c->stream));
    hipLaunchKernelGGL(k_site_pileup, dim3((uint32_t)((S.n + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, A);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->site_ev[1], c->stream));
    HIP_TRY(c, hipMemcpyAsync(hist, S.hist.p, n_sites * 16 * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    {
        float t = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&t, c->site_ev[0], c->site_ev[1]));
        c->site_ms = t;
        // 4-bit bases + per-read pos/mapq/offsets + CIGAR words read, the sites' positions /
        // indices read and their 16 counters written
        c->site_bytes = (S.nbase + 1) / 2 + S.n * sizeof(SiteRec) + S.ncig * 4 + (uint64_t)pos0.size() * 8 + (uint64_t)n_sites * 64;
    }
    tmr.lap("site run: kernel + histogram back");
    return CL_OK;
}

cl_status cl_site_pileup_stats(cl_ctx *c, double *kernel_ms, uint64_t *bytes)
{
    if (!c) return CL_ERR_INVALID;
    if (kernel_ms) *kernel_ms = c->site_ms;
    if (bytes) *bytes = c->site_bytes;
    return CL_OK;
}

cl_status cl_site_upload(cl_ctx *c, uint32_t contig_len, uint64_t ref_len, const cl_site_tile *t)
{
    // no exception leaves the library through the C ABI
    try { return cl_site_upload_impl(c, contig_len, ref_len, t); }
    catch (const std::bad_alloc &) { return fail(c, CL_ERR_NOMEM, "out of memory"); }
    catch (...) { return fail(c, CL_ERR_INVALID, "internal error"); }
}

cl_status cl_site_run(cl_ctx *c, uint8_t min_quality, const uint32_t *sites, size_t n_sites, uint32_t *hist)
{
    try { return cl_site_run_impl(c, min_quality, sites, n_sites, hist, nullptr); }
    catch (const std::bad_alloc &) { return fail(c, CL_ERR_NOMEM, "out of memory"); }
    catch (...) { return fail(c, CL_ERR_INVALID, "internal error"); }
}

cl_status cl_site_pileup(cl_ctx *c, uint8_t min_quality, uint32_t contig_len, uint64_t ref_len,
                         const cl_site_tile *t, const uint32_t *sites, size_t n_sites, uint32_t *hist)
{
    if (!c || !t || (!sites && n_sites) || (!hist && n_sites)) return CL_ERR_INVALID;
    if (n_sites == 0) return CL_OK;
    try {
        if (n_sites > 0x0FFFFFFFu) return fail(c, CL_ERR_RANGE, "too many sites");
        // the sorted site list first (a millisecond): it says which reads need to travel at all
        SitePrep prep;
        site_prepare(sites, n_sites, prep);
        SiteGather G;
        static const bool no_filter = [] { const char *e = getenv("DUT_SITE_FILTER"); return e && *e == '0'; }();   // =0: the whole tile travels (A/B, tests)
        if (!no_filter && t->n_reads && t->pos && t->mapq && t->cigar_off && t->seq_off) {
            StageTimer tf;
            site_filter(t, prep, min_quality, contig_len, G);
            tf.lap("site pileup: reads that overlap a site");
        }
        cl_status s = cl_site_upload_impl(c, contig_len, ref_len, t, &G);
        if (s != CL_OK) return s;
        return cl_site_run_impl(c, min_quality, sites, n_sites, hist, &prep);
    }
    catch (const std::bad_alloc &) { return fail(c, CL_ERR_NOMEM, "out of memory"); }
    catch (...) { return fail(c, CL_ERR_INVALID, "internal error"); }
}


} // extern "C"
