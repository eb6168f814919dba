// tf_device.hip -- context, device memory, launch logic and the C ABI of libtunafock.so (include/tunafock.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <array>
#include <cstring>
#include <condition_variable>
#include <dlfcn.h>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/tunafock.h"
#include "tf_internal.h"
#include "tf_kernels.hip.h"
#include "tf_jkpacked.hip.h"
#include "tf_tiles_host.h"
#include "tf_jktile.hip.h"
#include "tf_eri.hip.h"
#include "tf_eri_team.hip.h"
#include "tf_eri_teamc.hip.h"
#include "tf_eri_team_api.h"
#include "tf_oneel.hip.h"
#include "tf_scf.hip.h"
#include "tf_mp2.hip.h"
#include "tf_dft.hip.h"

// ---- small-block cache for THIS file's device allocations ------------------------------------------------------------------------
// A tensor build uploads ~40 small tables (layout, rows, work lists) and frees them again; every hipMalloc / hipFree pair costs 20-40 us
// of host time and hipFree synchronises the device -- at N = 60 that is a third of the 3.3 ms a build takes.  Blocks of up to 64 MiB are
// therefore recycled: size classes of powers of two (>= 512 bytes), at most 1 GiB kept; larger blocks and unknown pointers go straight
// to the runtime.  What hipFree's implicit synchronisation used to guarantee -- no kernel still reads a block that is handed out
// again -- is kept by the explicit hipDeviceSynchronize() in free_eri / free_basis (the two places that free buffers kernels of
// EARLIER calls may still use).  Per process and device, mutex-protected (the lockstep SCF batch runs on several host threads).
namespace tfcache {
struct Block { void *p; int dev; };
static std::mutex mu;
static std::multimap<size_t, Block> free_blocks;            // size class -> cached block
static std::map<void *, std::pair<size_t, int>> live;       // blocks handed out by cmalloc: (size class, device)
static size_t cached_bytes = 0;
static const size_t MAX_BLOCK = (size_t)64 << 20, MAX_CACHED = (size_t)1 << 30;
static const bool off = getenv("TF_ALLOC_CACHE") && getenv("TF_ALLOC_CACHE")[0] == '0';
static size_t size_class(size_t b) { size_t c = 512; while (c < b) c <<= 1; return c; }
// Hands every cached block back to the runtime (after a device-wide synchronisation: a kernel of an earlier call may still read one).
// Called when the runtime is out of memory -- the cache must never be the reason a build fails -- and by tf_destroy of the last context.
static void trim()
{
    std::lock_guard<std::mutex> lk(mu);
    if (free_blocks.empty()) return;
    int dev0 = 0;
    (void)hipGetDevice(&dev0);
    for (auto &kv : free_blocks) {
        (void)hipSetDevice(kv.second.dev);
        (void)hipDeviceSynchronize();
        (void)::hipFree(kv.second.p);
    }
    (void)hipSetDevice(dev0);
    free_blocks.clear();
    cached_bytes = 0;
}
static size_t cached() { std::lock_guard<std::mutex> lk(mu); return cached_bytes; }
static hipError_t raw_malloc(void **p, size_t bytes)
{
    hipError_t e = ::hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory && cached() > 0) {
        (void)hipGetLastError();
        trim();
        e = ::hipMalloc(p, bytes);
    }
    return e;
}
static hipError_t cmalloc(void **p, size_t bytes)
{
    if (off || bytes > MAX_BLOCK) return raw_malloc(p, bytes);
    int dev = 0;
    (void)hipGetDevice(&dev);
    const size_t c = size_class(bytes);
    {
        std::lock_guard<std::mutex> lk(mu);
        auto range = free_blocks.equal_range(c);
        for (auto it = range.first; it != range.second; ++it)
            if (it->second.dev == dev) {
                *p = it->second.p;
                free_blocks.erase(it);
                cached_bytes -= c;
                live[*p] = std::make_pair(c, dev);
                return hipSuccess;
            }
    }
    hipError_t e = raw_malloc(p, c);
    if (e == hipSuccess) { std::lock_guard<std::mutex> lk(mu); live[*p] = std::make_pair(c, dev); }
    return e;
}
static thread_local bool bypass = false;     // set while blocks are released after a failed synchronisation: straight back to the runtime
static hipError_t cfree(void *p)
{
    if (!p) return hipSuccess;
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = live.find(p);
        if (it != live.end()) {
            const size_t c = it->second.first;
            const int dev = it->second.second;
            live.erase(it);
            if (!bypass && cached_bytes + c <= MAX_CACHED) {
                free_blocks.emplace(c, Block{p, dev});
                cached_bytes += c;
                return hipSuccess;
            }
        }
    }
    return ::hipFree(p);
}
}  // namespace tfcache
// every device allocation of this file goes through the cache (blocks released elsewhere with the runtime's hipFree use ::hipMalloc)
static inline hipError_t tf_malloc(void **p, size_t bytes) { return tfcache::cmalloc(p, bytes); }
template <class T> static inline hipError_t tf_malloc(T **p, size_t bytes) { return tfcache::cmalloc((void **)p, bytes); }
static inline hipError_t tf_free(void *p) { return tfcache::cfree(p); }

using namespace tfk;

static std::string g_create_error;
static std::atomic<int> g_live_contexts{0};
static const bool g_dbg = getenv("TF_DEBUG") != nullptr;
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define DBG(...) do { if (g_dbg) { fprintf(stderr, "[tf %.6f] ", now_s()); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); fflush(stderr); } } while (0)

struct tf_ctx {
    int device = 0, rank = 0, world = 1;
    mutable std::string err;
    tf::Basis bs;
    bool have_basis = false, have_eri = false;
    // device copy of the basis
    DBasis db{};
    std::vector<void *> basis_allocs;
    int *d_csr_ptr = nullptr, *d_csr_idx = nullptr;
    double *d_csr_val = nullptr;
    int csr_rows = 0;
    // stored tensor
    int spherical = 1, N = 0, ld = 0;
    long long n_rows = 0;
    double *d_eri = nullptr;
    size_t eri_cap = 0;                  // bytes allocated behind d_eri: kept across builds (freeing and reallocating 27 GB costs ~1.4 s)
    int2 *d_row_ij = nullptr;
    int *d_class_rows = nullptr, *d_row_pos = nullptr;   // packed layout: local rows listed class by class; position of a row in that order
    long long class_row_off[5] = {0, 0, 0, 0, 0};         // rows of class c: d_class_rows[class_row_off[c] .. class_row_off[c + 1])
    int *d_rowmap = nullptr;
    std::vector<int> my_pairs;          // bra shell pairs owned by this rank
    std::vector<DPair> host_pairs;      // host mirror of db.pairs (output offsets are filled in by tf_build_eri)
    DPair *d_pairs = nullptr;
    std::vector<int> pair_class;        // class id of every shell pair
    std::vector<int> h_ct_ix, h_ct_ord;  // host mirrors of DBasis::ct_ix / ct_ord / ct_sc (the team kernels' transform tables are built from them)
    std::vector<double> h_ct_sc;
    std::vector<std::vector<int>> class_pairs;   // pairs of each class, ascending
    // J/K scratch
    double *d_Jrow = nullptr, *d_Kp = nullptr, *d_Ppad = nullptr, *d_J = nullptr, *d_K = nullptr, *d_P = nullptr;
    // packed (8-fold unique) layout: tf_jkpacked.hip.h
    int layout_req = -1;                // -1 auto, 0 rows (i >= j) x [k][l], 1 packed
    int layout = 0;
    long long n_elems = 0;              // stored doubles
    long long *d_rowoff = nullptr;
    int *d_jptr = nullptr;                       // rows by second index x (internal): d_jrows[d_jptr[x] .. d_jptr[x + 1])
    int2 *d_jrows = nullptr;                     // (local row, ORIGINAL first index of the row)
    int *d_xorder = nullptr;                     // output rows of the exchange reduction, most partial vectors first
    int *d_rowsec = nullptr;            // [n_rows][6]: start of section a inside local row r; position in its storage unit, rows of the unit
    // parity-blocked layout tables (tf_layout.hip.h), host mirror and device view
    struct HostLayout {
        int N = 0, NW = 0, RS = 0, MC = 1;
        int KS = 1 << 30, MP = 1;                                  // steps per part and parts of a cut walk (several ranks: shorter tasks)
        int cstart[4] = {}, csize[4] = {}, corder[4] = {}, wfirst[5] = {}, fullsec[4][4] = {}, gbase[4] = {};
        long long cbase[4] = {}, NP[4] = {}, NPtot = 0, RLS = 0;
        std::vector<int> cls, loc, sigma, ao, origI, clsI, cntA, kap0, kapF, rpoff, chunk_c0, chunk_width, chunk_cls, chunk_of, gk;
        std::vector<KInfo> kinfo;                                   // [4][N]
        std::vector<int> offE;                                      // [4][N]: offA + padded segment length
        int ke(int a, int xI) const { return cntA[(size_t)a * N + xI]; }
        int seclen(int c, int a, int iI) const { const int k = ke(a, iI); return k == 0 ? 0 : offE[(size_t)c * N + cstart[a] + k - 1]; }
        int row_shape(int c, int iI, int *secoff) const {           // section starts and the length of a class-c row with first index iI
            int tot = 0;
            for (int t = 0; t < 4; ++t) { const int a = corder[t]; secoff[a] = tot; tot += seclen(c, a, iI); }
            return tot;
        }
        bool task_exists(int c, int w, int iI) const { return kap0[(size_t)c * NW + w] < ke(chunk_cls[w] ^ c, iI); }
    } hl;
    BLayout bl{};
    std::vector<void *> layout_allocs;
    // work tables of jk_packed_kernel: set 0 for one density per pass (groups of 8 rows), set 1 for two (groups of 4 rows)
    struct JKTables {
        JKGroup *d_groups = nullptr;
        JKTask *d_tasks = nullptr;
        JKSuper *d_supers = nullptr;
        int *d_gfirst = nullptr;        // [2][N]: first / one-past-last group with i == a
        int n_groups = 0, n_tasks = 0, n_supers = 0, nseg = 1;
        int bucket[4] = {0, 0, 0, 0};   // tasks [bucket[b], bucket[b + 1]) run with 4, 2, 1 waves per workgroup (b = 0, 1, 2)
        // the tasks a CLASS-DIAGONAL density needs (same order, same buckets): a row (i, j) of class c != 0 only meets P[j][l], P[j][k],
        // P[i][k], P[i][l] and the pair densities of class 0 -- every product of a task whose column class is neither i's nor j's is zero
        JKTask *d_tasks_cd = nullptr;
        int n_tasks_cd = 0, bucket_cd[4] = {0, 0, 0, 0};
        long long ypart_len = 0;
        JKJtPlan jp{};
    } jkt[2];
    double *d_Psym = nullptr, *d_Pp = nullptr, *d_ypart = nullptr, *d_DI = nullptr, *d_DJ = nullptr, *d_Jt = nullptr, *d_D = nullptr;
    // second set of partial-sum buffers for passes over the class-diagonal task list (the entries its skipped tasks would write stay
    // zero, which the reductions rely on); allocated at the first such pass.  cd_bytes: sizes of d_Jrow, d_ypart, d_DI, d_DJ
    double *cd_Jrow = nullptr, *cd_ypart = nullptr, *cd_DI = nullptr, *cd_DJ = nullptr;
    size_t cd_bytes[4] = {0, 0, 0, 0};
    unsigned long long *d_cdflag = nullptr;       // device word: largest |P| between AOs of different classes (bit pattern), per build
    bool jk_try_class_diagonal = false;           // set by the SCF cycles around their Fock builds (their densities usually are)
    long long jk_cd_passes = 0, jk_cd_declined = 0;
    // tiles layout (tf_tiles.h, tf_jktile.hip.h): host tables, their device copies and the partial-sum buffers of the Fock kernel
    tft::Tables tiles;
    tft::TaskList tsub1;                 // the one-density task list when its strips are shorter than the stored ones (TF_TILE_KSUB)
    int ksub1 = TT_KS;
    TView tv{};
    struct TileList {                    // device copy of a tft::TaskList + the partial-sum buffers of its passes
        TTask *d_tasks = nullptr;
        TPairI *d_pairs = nullptr;
        TRunI *d_runs = nullptr;
        int *d_itask_ptr = nullptr, *d_itasks = nullptr;
        int n_tasks = 0, bucket[TT_W + 1] = {0, 0, 0, 0, 0}, ksub = TT_KS, n_di = 0;
        long long dj_len = 0, jd_len = 0, jt_len = 0;
        double *DJ = nullptr, *Jt = nullptr, *Jd = nullptr, *DIk = nullptr, *DIl = nullptr;
        int nd_cap = 0;                  // densities the buffers are sized for
    } tl1, tlw;                          // one density per pass (strips of ksub1 rows); 4 / 8 densities per pass (strips of 16 rows: built at first use)
    tft::TaskList tsubw;
    tft::ClassInfo tclass;               // what the list builder needs again for tlw
    std::vector<std::pair<int, int>> trows;
    int *d_jlist_ptr = nullptr, *d_jlist = nullptr, *d_tvtab = nullptr;
    double *t_X = nullptr, *t_Pm = nullptr;   // [8][N][N] internal densities, [8] pair matrices
    double *t_out = nullptr;             // [6: Dj, Di, ED, EDT, JD, EJ][densities of the pass][N][N]; two such sets (the two passes of a general density)
    double *t_JtTot = nullptr;
    std::vector<void *> tile_allocs;
    size_t tile_lds_set = 48 * 1024;     // dynamic LDS limit requested for the tiles layout's edge / reduce kernels
    // instrumentation
    bool prof_jk = false;
    std::vector<hipEvent_t> prof_ev;     // pairs (before, after) around the row kernel, on the launch stream
    size_t prof_used = 0;
    void *comm = nullptr;                 // RCCL communicator of the ranks that share the tensor (tf_comm_init): the exchange step without a host callback
    tf_allreduce_fn allreduce = nullptr;  // completes partial [J;K] over the ranks of a sharded tensor (tf_set_allreduce)
    void *allreduce_user = nullptr;
    double *d_jkstage = nullptr;          // [2][nd][N][N] staging buffer of that exchange step
    double *d_agree = nullptr;            // 16 doubles: per-iteration decision values summed over the ranks (agree_over_ranks)
    size_t jkstage_doubles = 0;
    double eri_seconds[4] = {0, 0, 0, 0};
    long long eri_counts[3] = {0, 0, 0};
    double eri_nominal_flops = 0.0;      // the reference algorithm's operation count for the quartets of the last build
    tfscf::Workspace scf;
    std::vector<std::unique_ptr<tfscf::Workspace>> scf_batch;   // one workspace per cycle of a lockstep batch (tf_scf_rhf_batch), kept across calls
    tfdft::Grid grid;                    // Kohn-Sham integration grid with the AOs evaluated on it (tf_dft_setup)
    // persistent helpers of tf_build_eri (creating streams / freeing GiB-sized buffers costs tens of ms per call)
    static const int NSTREAM_MAX = 8;
    hipStream_t streams[NSTREAM_MAX] = {};
    hipEvent_t sev[NSTREAM_MAX] = {};
    bool have_streams = false;
    hipStream_t cstream = nullptr;            // uploads of a slab's index lists (beside the kernels of the previous slab)
    hipEvent_t slab_done[2] = {nullptr, nullptr}, slab_lists[2] = {nullptr, nullptr};
    size_t cfact_lds_set = 0;                // dynamic LDS limit requested for eri_cfact_kernel
    tfk::LRec *d_lrec = nullptr;             // per-(La,Lb|Lc,Ld) records and entry index words of eri_cfact_kernel (per build)
    unsigned short *d_tup = nullptr;
    tfone::Arena arena1e;                    // device block for the short-lived buffers of the one-electron integrals
    double *d_gtab = nullptr;                // global-memory tables of the top angular momenta (eri_cfact_kernel<true, true>)
    size_t gtab_bytes = 0;
    double *scr[3] = {nullptr, nullptr, nullptr};
    size_t scr_bytes[3] = {0, 0, 0};
    double *mo_pool = nullptr;               // work space of the short-index-first AO->MO transformation (tfmp2::transform_q1), kept across calls
    size_t mo_pool_bytes = 0;
};

static int ensure_scratch(tf_ctx *ctx, int k, size_t bytes)
{
    if (bytes <= ctx->scr_bytes[k]) return TF_OK;
    if (ctx->scr[k]) { (void)tf_free(ctx->scr[k]); ctx->scr[k] = nullptr; ctx->scr_bytes[k] = 0; }
    hipError_t e = tf_malloc((void **)&ctx->scr[k], bytes);
    if (e != hipSuccess) { ctx->err = std::string("hipMalloc of ERI scratch failed: ") + hipGetErrorString(e); return TF_ENOMEM; }
    ctx->scr_bytes[k] = bytes;
    return TF_OK;
}

#define TF_FAIL(ctx, code, ...)                                  \
    do {                                                         \
        char _b[512];                                            \
        snprintf(_b, sizeof(_b), __VA_ARGS__);                   \
        (ctx)->err = _b;                                         \
        return (code);                                           \
    } while (0)

#define HIPCHK(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t _e = (call);                                                                    \
        if (_e != hipSuccess)                                                                      \
            TF_FAIL(ctx, (_e == hipErrorOutOfMemory ? TF_ENOMEM : TF_ENODEVICE), "%s failed: %s (%s:%d)", #call, \
                    hipGetErrorString(_e), __FILE__, __LINE__);                                    \
    } while (0)

template <class T>
static int upload(tf_ctx *ctx, const std::vector<T> &h, T **d, bool track = true)
{
    size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(T);
    HIPCHK(ctx, tf_malloc((void **)d, bytes));
    if (track) ctx->basis_allocs.push_back(*d);
    if (!h.empty()) HIPCHK(ctx, hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return TF_OK;
}

// ---- RCCL, loaded on demand (include/tunafock.h: tf_comm_*) ------------------------------------------------------------------------
namespace tfrccl {
typedef struct { char internal[128]; } UniqueId;                // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef int (*GetUniqueId_t)(UniqueId *);
typedef int (*CommInitRank_t)(void **, int, UniqueId, int);
typedef int (*CommDestroy_t)(void *);
typedef int (*AllReduce_t)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef int (*Group_t)(void);
typedef const char *(*GetErrorString_t)(int);
static std::mutex mu;
static void *handle = nullptr;
static GetUniqueId_t GetUniqueId = nullptr;
static CommInitRank_t CommInitRank = nullptr;
static CommDestroy_t CommDestroy = nullptr;
static AllReduce_t AllReduce = nullptr;
static Group_t GroupStart = nullptr, GroupEnd = nullptr;
static GetErrorString_t GetErrorString = nullptr;
enum { kFloat64 = 8, kSum = 0 };                                   // ncclFloat64, ncclSum
static bool load(std::string &err)
{
    std::lock_guard<std::mutex> lk(mu);
    if (handle) return true;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (handle) break;
    }
    if (!handle) { err = std::string("librccl not found: ") + (dlerror() ? dlerror() : ""); return false; }
    GetUniqueId = (GetUniqueId_t)dlsym(handle, "ncclGetUniqueId"); CommInitRank = (CommInitRank_t)dlsym(handle, "ncclCommInitRank");
    CommDestroy = (CommDestroy_t)dlsym(handle, "ncclCommDestroy"); AllReduce = (AllReduce_t)dlsym(handle, "ncclAllReduce");
    GroupStart = (Group_t)dlsym(handle, "ncclGroupStart"); GroupEnd = (Group_t)dlsym(handle, "ncclGroupEnd");
    GetErrorString = (GetErrorString_t)dlsym(handle, "ncclGetErrorString");
    if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllReduce || !GroupStart || !GroupEnd) { err = "librccl lacks an expected symbol"; dlclose(handle); handle = nullptr; return false; }
    return true;
}
static std::string errstr(int code) { return GetErrorString ? std::string(GetErrorString(code)) : ("code " + std::to_string(code)); }
}  // namespace tfrccl

// sum over the ranks of the attached communicator, in place, on stream st
static int comm_allreduce(tf_ctx *ctx, double *buf, size_t count, hipStream_t st)
{
    const int rc = tfrccl::AllReduce(buf, buf, count, tfrccl::kFloat64, tfrccl::kSum, ctx->comm, st);
    if (rc) TF_FAIL(ctx, TF_ENODEVICE, "ncclAllReduce failed: %s", tfrccl::errstr(rc).c_str());
    return TF_OK;
}

extern "C++" {
static int tile_list_upload(tf_ctx *ctx, const tft::TaskList &TL, bool own_shapes, tf_ctx::TileList &D);
}

// The blocks freed by free_eri / free_basis are recycled at once (tfcache): nothing queued earlier may still use them, hence the
// synchronisation.  If it FAILS the device is in an error state: the error is kept for the caller (tf_last_error), the cache is emptied
// and the blocks of this call go straight back to the runtime instead of being handed out again.
struct DrainedFree {
    bool ok;
    DrainedFree(tf_ctx *ctx, const char *what)
    {
        const hipError_t e = hipDeviceSynchronize();
        ok = e == hipSuccess;
        if (!ok) {
            ctx->err = std::string(what) + ": hipDeviceSynchronize failed before buffers were released (" + hipGetErrorString(e) + ")";
            tfcache::trim();
            tfcache::bypass = true;
        }
    }
    ~DrainedFree() { tfcache::bypass = false; }
};

static void free_eri(tf_ctx *ctx)
{
    DrainedFree drained(ctx, "free_eri");
    for (void *p : {(void *)ctx->d_class_rows, (void *)ctx->d_row_pos, (void *)ctx->d_row_ij, (void *)ctx->d_rowmap, (void *)ctx->d_Jrow, (void *)ctx->d_Kp,
                    (void *)ctx->d_Ppad, (void *)ctx->d_J, (void *)ctx->d_K, (void *)ctx->d_P, (void *)ctx->d_rowoff, (void *)ctx->d_Psym,
                    (void *)ctx->d_Pp, (void *)ctx->d_ypart, (void *)ctx->d_DI, (void *)ctx->d_DJ, (void *)ctx->d_Jt, (void *)ctx->d_D})
        if (p) (void)tf_free(p);
    for (void *p : {(void *)ctx->cd_Jrow, (void *)ctx->cd_ypart, (void *)ctx->cd_DI, (void *)ctx->cd_DJ, (void *)ctx->d_cdflag})
        if (p) (void)tf_free(p);
    ctx->cd_Jrow = ctx->cd_ypart = ctx->cd_DI = ctx->cd_DJ = nullptr; ctx->d_cdflag = nullptr;
    for (auto &t : ctx->jkt) {
        for (void *p : {(void *)t.d_groups, (void *)t.d_tasks, (void *)t.d_supers, (void *)t.d_gfirst, (void *)t.d_tasks_cd})
            if (p) (void)tf_free(p);
        t = tf_ctx::JKTables();
    }
    for (void *p : {(void *)ctx->d_jptr, (void *)ctx->d_jrows, (void *)ctx->d_xorder})
        if (p) (void)tf_free(p);
    ctx->d_jptr = nullptr; ctx->d_jrows = nullptr; ctx->d_xorder = nullptr;
    for (void *p : ctx->layout_allocs) (void)tf_free(p);
    ctx->layout_allocs.clear();
    for (void *p : ctx->tile_allocs) (void)tf_free(p);
    ctx->tile_allocs.clear();
    ctx->tl1 = tf_ctx::TileList(); ctx->tlw = tf_ctx::TileList();
    ctx->tv = TView{};
    ctx->d_jlist_ptr = ctx->d_jlist = ctx->d_tvtab = nullptr;
    ctx->t_X = ctx->t_Pm = ctx->t_out = ctx->t_JtTot = nullptr;
    if (ctx->d_rowsec) { (void)tf_free(ctx->d_rowsec); ctx->d_rowsec = nullptr; }
    ctx->bl = BLayout{};
    ctx->d_class_rows = nullptr; ctx->d_row_pos = nullptr; ctx->d_row_ij = nullptr; ctx->d_rowmap = nullptr; ctx->d_Jrow = nullptr; ctx->d_Kp = nullptr;
    ctx->d_Ppad = nullptr; ctx->d_J = nullptr; ctx->d_K = nullptr; ctx->d_P = nullptr;
    ctx->d_rowoff = nullptr; ctx->d_Psym = nullptr; ctx->d_Pp = nullptr;
    ctx->d_ypart = nullptr; ctx->d_DI = nullptr; ctx->d_DJ = nullptr; ctx->d_Jt = nullptr; ctx->d_D = nullptr;
    ctx->n_elems = 0;
    ctx->have_eri = false;
}

static void free_basis(tf_ctx *ctx)
{
    DrainedFree drained(ctx, "free_basis");
    for (void *p : ctx->basis_allocs) (void)tf_free(p);
    ctx->basis_allocs.clear();
    for (void *p : {(void *)ctx->d_csr_ptr, (void *)ctx->d_csr_idx, (void *)ctx->d_csr_val})
        if (p) (void)tf_free(p);
    ctx->d_csr_ptr = ctx->d_csr_idx = nullptr; ctx->d_csr_val = nullptr;
    ctx->have_basis = false;
}

// Tables of the parity-blocked layout for the output AOs of this build (tf_layout.hip.h; the NumPy model tests/layout_model.py
// builds the same tables).  cls[k]: x/y parity class of output AO k (original order).
static int build_blocked_layout(tf_ctx *ctx, const std::vector<int> &cls)
{
    tf_ctx::HostLayout &H = ctx->hl;
    H = tf_ctx::HostLayout();
    const int N = (int)cls.size(), PAD = TF_SEG_PAD;
    H.N = N; H.cls = cls;
    for (int k = 0; k < N; ++k) ++H.csize[cls[k]];
    int order[4] = {0, 1, 2, 3};
    std::stable_sort(order, order + 4, [&](int x, int y) { return H.csize[x] > H.csize[y]; });   // larger classes first (ties: class id)
    for (int t = 0, s0 = 0; t < 4; ++t) { H.corder[t] = order[t]; H.cstart[order[t]] = s0; s0 += H.csize[order[t]]; }
    H.loc.assign(N, 0); H.sigma.assign(N, 0); H.ao.assign(N, 0); H.origI.assign(N, 0); H.clsI.assign(N, 0);
    std::vector<int> cnt((size_t)4 * N, 0);                          // cnt[b][k]: class-b AOs with original index <= k
    {
        int seen[4] = {0, 0, 0, 0};
        for (int k = 0; k < N; ++k) {
            H.loc[k] = seen[cls[k]]++;
            H.sigma[k] = H.cstart[cls[k]] + H.loc[k];
            H.ao[k] = cls[k] | (H.loc[k] << 2);
            H.origI[H.sigma[k]] = k;
            H.clsI[H.sigma[k]] = cls[k];
            for (int b = 0; b < 4; ++b) cnt[(size_t)b * N + k] = seen[b];
        }
    }
    H.cntA.assign((size_t)4 * N, 0);
    for (int a = 0; a < 4; ++a)
        for (int x = 0; x < N; ++x) H.cntA[(size_t)a * N + x] = cnt[(size_t)a * N + H.origI[x]];
    H.kinfo.assign((size_t)4 * N, KInfo{0, 0});
    H.offE.assign((size_t)4 * N, 0);
    for (int c = 0; c < 4; ++c) {
        long long tot = 0;
        for (int t = 0; t < 4; ++t) {
            const int a = H.corder[t];
            H.fullsec[c][a] = (int)tot;
            long long off = 0;
            for (int kk = 0; kk < H.csize[a]; ++kk) {
                const int kI = H.cstart[a] + kk;
                const int n = cnt[(size_t)(a ^ c) * N + H.origI[kI]];
                H.kinfo[(size_t)c * N + kI] = KInfo{(int)off, n};
                off += (n + PAD - 1) / PAD * PAD;
                H.offE[(size_t)c * N + kI] = (int)off;
            }
            tot += off;
        }
        if (tot > 0x7fffffffLL / 8) TF_FAIL(ctx, TF_EINVAL, "basis too large for the packed layout's 32-bit row offsets");
        H.NP[c] = tot;
    }
    H.NPtot = 0; H.RLS = 0;
    for (int c = 0; c < 4; ++c) { H.cbase[c] = H.NPtot; H.NPtot += H.NP[c]; H.RLS = std::max(H.RLS, H.NP[c]); }
    // granule table: AO k of the segment that holds granule g of class c's pair index space
    {
        int gb = 0;
        for (int c = 0; c < 4; ++c) { H.gbase[c] = gb; gb += (int)(H.NP[c] / PAD); }
        H.gk.assign((size_t)std::max(gb, 1), 0);
        for (int c = 0; c < 4; ++c)
            for (int kI = 0; kI < N; ++kI) {
                const int a = H.clsI[kI];
                const int g0 = (H.fullsec[c][a] + H.kinfo[(size_t)c * N + kI].offA) / PAD, g1 = (H.fullsec[c][a] + H.offE[(size_t)c * N + kI]) / PAD;
                for (int g = g0; g < g1; ++g) H.gk[(size_t)H.gbase[c] + g] = kI;
            }
    }
    // column chunks: the internal columns cut at class boundaries and every TF_JKP_CW columns
    H.chunk_of.assign(N, 0);
    for (int b = 0; b < 4; ++b) {
        H.wfirst[b] = (int)H.chunk_cls.size();
        for (int lam0 = 0; lam0 < H.csize[b]; lam0 += TF_JKP_CW) {
            const int wd = std::min(TF_JKP_CW, H.csize[b] - lam0);
            for (int u = 0; u < wd; ++u) H.chunk_of[H.cstart[b] + lam0 + u] = (int)H.chunk_cls.size();
            H.chunk_cls.push_back(b); H.chunk_c0.push_back(H.cstart[b] + lam0); H.chunk_width.push_back(wd);
        }
    }
    H.wfirst[4] = (int)H.chunk_cls.size();
    H.NW = (int)H.chunk_cls.size();
    const int NW = H.NW;
    H.kap0.assign((size_t)4 * std::max(NW, 1), 0); H.kapF.assign((size_t)4 * std::max(NW, 1), 0); H.rpoff.assign((size_t)4 * std::max(NW, 1), 0);
    // row parts of a group / row: a dense [MC][N] block, MC = most chunks of one class; the task of chunk number s of its class writes
    // slot s at the internal index of k: rpoff[c][w] = s N + cstart[class of k] (+ kappa)
    H.MC = 1;
    for (int b = 0; b < 4; ++b) H.MC = std::max(H.MC, H.wfirst[b + 1] - H.wfirst[b]);
    H.RS = H.MC * N;
    {
        // Several ranks: a rank has 1/world of the tasks but every task walks as long as before, so the longest walks bound the pass
        // (N = 400, 8 ranks: 0.58 ms against 0.24 ms at perfect balance).  The walks are cut into MP parts of KS steps; the price is one
        // plane of column parts (and of Jd) per part.
        int parts = ctx->world >= 4 ? 4 : (ctx->world >= 2 ? 2 : 1);
        if (const char *e = getenv("TF_JK_PARTS")) parts = std::max(1, std::min(8, atoi(e)));
        int longest = 1;
        for (int a = 0; a < 4; ++a) longest = std::max(longest, H.csize[a]);
        H.MP = std::max(1, std::min(parts, longest));
        H.KS = (longest + H.MP - 1) / H.MP;
    }
    for (int c = 0; c < 4; ++c) {
        for (int w = 0; w < NW; ++w) {
            const int b = H.chunk_cls[w], a = b ^ c, lam0 = H.chunk_c0[w] - H.cstart[b];
            const int cm = (c == 0) ? 1 : 0;
            int k0 = H.csize[a], kF = H.csize[a];
            for (int kk = H.csize[a] - 1; kk >= 0; --kk) {           // the counts are non-decreasing along a class
                const int n = H.kinfo[(size_t)c * N + H.cstart[a] + kk].cnt;
                if (n > lam0) k0 = kk;
                if (H.chunk_width[w] == TF_JKP_CW && n - cm >= lam0 + TF_JKP_CW) kF = kk;
            }
            H.kap0[(size_t)c * NW + w] = k0; H.kapF[(size_t)c * NW + w] = kF;
            H.rpoff[(size_t)c * NW + w] = (w - H.wfirst[b]) * N + H.cstart[a];
        }
    }
    // device copy
    BLayout L{};
    L.N = N; L.NW = NW; L.RS = H.RS; L.NPtot = H.NPtot;
    std::vector<int> itab(BL_ITAB, 0);
    std::vector<long long> ltab(BL_LTAB, 0);
    for (int c = 0; c < 4; ++c) {
        itab[BL_CSTART + c] = H.cstart[c]; itab[BL_CSIZE + c] = H.csize[c]; itab[BL_GBASE + c] = H.gbase[c];
        ltab[BL_CBASE + c] = H.cbase[c]; ltab[BL_NP + c] = H.NP[c];
        for (int a = 0; a < 4; ++a) itab[BL_FULLSEC + 4 * c + a] = H.fullsec[c][a];
    }
    for (int b = 0; b < 5; ++b) itab[BL_WFIRST + b] = H.wfirst[b];
    auto up_i = [&](const std::vector<int> &h, const int **d) -> int {
        int *p = nullptr;
        int rc = upload(ctx, h, &p, false);
        if (rc) return rc;
        ctx->layout_allocs.push_back(p);
        *d = p;
        return TF_OK;
    };
    int rc;
    {
        long long *dl = nullptr;
        if ((rc = upload(ctx, ltab, &dl, false))) return rc;
        ctx->layout_allocs.push_back(dl);
        L.ltab = dl;
    }
    if ((rc = up_i(itab, &L.itab)) || (rc = up_i(H.ao, &L.ao)) || (rc = up_i(H.origI, &L.origI)) || (rc = up_i(H.clsI, &L.clsI)) || (rc = up_i(H.cntA, &L.cntA)) ||
        (rc = up_i(H.kap0, &L.kap0)) || (rc = up_i(H.kapF, &L.kapF)) || (rc = up_i(H.rpoff, &L.rpoff)) || (rc = up_i(H.chunk_c0, &L.chunk_c0)) ||
        (rc = up_i(H.chunk_width, &L.chunk_width)) || (rc = up_i(H.chunk_cls, &L.chunk_cls)) || (rc = up_i(H.chunk_of, &L.chunk_of)) ||
        (rc = up_i(H.gk, &L.gk)))
        return rc;
    KInfo *dk = nullptr;
    if ((rc = upload(ctx, H.kinfo, &dk, false))) return rc;
    ctx->layout_allocs.push_back(dk);
    L.kinfo = dk;
    ctx->bl = L;
    return TF_OK;
}

// Longest-processing-time assignment of row blocks (bra shell pairs) to ranks: heaviest block first, always to
// the least loaded rank.  Deterministic, so every rank computes the same plan without communication.
static void shard_plan(const std::vector<long long> &weight, int world, std::vector<int> &owner)
{
    const int n = (int)weight.size();
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return weight[x] > weight[y]; });
    std::vector<long long> load(world, 0);
    owner.assign(n, 0);
    for (int p : order) {
        const int r = (int)(std::min_element(load.begin(), load.end()) - load.begin());
        load[r] += weight[p];
        owner[p] = r;
    }
}

// The plan of tf_build_eri (include/tunafock.h: tf_shard_plan_pairs).  Deterministic: every rank computes the same plan.
static void shard_plan_pairs(const std::vector<int> &dim, bool packed, int world, std::vector<int> &owner)
{
    const int ns = (int)dim.size();
    std::vector<long long> off(ns + 1, 0);
    for (int s = 0; s < ns; ++s) off[s + 1] = off[s] + dim[s];
    owner.assign((size_t)ns * (ns + 1) / 2, 0);
    std::vector<long long> load(world, 0);
    std::vector<long long> w;
    // WHOLE bra shells per rank when that balances (round 3): a rank then holds complete runs of j for its rows i -- super-groups as full
    // as on one GPU, and an eighth of the super-groups, so that the Jt partials and their reduction shrink with the shard.  (With every A
    // cut into `world` segments each rank keeps a short run of j under EVERY i: at N = 400 on 8 ranks the per-rank reduction stayed at
    // 0.08 ms -- as many super-groups per rank as on one GPU -- and the J/K kernel ran on one-group workgroups.)  Longest-processing-time
    // over the shells' weights; taken if the heaviest rank is within 3 % of the mean, else the segment plan below.  TF_SHARD_PLAN=segments
    // / shells forces one of the two (identically on every rank, of course).
    {
        const char *pe = getenv("TF_SHARD_PLAN");
        const bool force_seg = pe && pe[0] == 's' && pe[1] == 'e', force_sh = pe && pe[0] == 's' && pe[1] == 'h';
        if (world > 1 && !force_seg) {
            std::vector<long long> wA(ns, 0);
            long long total = 0;
            for (int A = 0; A < ns; ++A) {
                for (long long i = off[A]; i < off[A + 1]; ++i)
                    for (long long j = 0; j <= i; ++j) wA[A] += packed ? packed_row_len(i, j) : 1;
                total += wA[A];
            }
            std::vector<int> ord(ns);
            std::iota(ord.begin(), ord.end(), 0);
            std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return wA[x] > wA[y]; });
            std::vector<long long> ld(world, 0);
            std::vector<int> rankA(ns, 0);
            for (int A : ord) {
                const int r = (int)(std::min_element(ld.begin(), ld.end()) - ld.begin());
                ld[r] += wA[A];
                rankA[A] = r;
            }
            const long long heaviest = *std::max_element(ld.begin(), ld.end());
            if (force_sh || (double)heaviest * world <= 1.03 * (double)total) {
                for (int A = 0; A < ns; ++A)
                    for (int B = 0; B <= A; ++B) owner[(size_t)A * (A + 1) / 2 + B] = rankA[A];
                return;
            }
        }
    }
    for (int A = ns - 1; A >= 0; --A) {                           // heaviest rows first
        w.assign(A + 1, 0);
        long long total = 0;
        for (int B = 0; B <= A; ++B) {
            long long wb = 0;
            for (long long i = off[A]; i < off[A + 1]; ++i)
                for (long long j = off[B]; j < off[B + 1] && j <= i; ++j) wb += packed ? packed_row_len(i, j) : 1;
            w[B] = wb; total += wb;
        }
        // `world` contiguous segments of B with (nearly) equal weight
        struct Seg { int b0, b1; long long wt; };
        std::vector<Seg> segs;
        int b = 0;
        long long cum = 0;
        for (int sgi = 1; sgi <= world; ++sgi) {
            const long long target = (long long)((__int128)total * sgi / world);
            Seg sg{b, b, 0};
            while (b <= A && (sgi == world || cum + w[b] / 2 <= target)) { cum += w[b]; sg.wt += w[b]; ++b; }
            sg.b1 = b;
            segs.push_back(sg);
        }
        std::stable_sort(segs.begin(), segs.end(), [](const Seg &x, const Seg &y) { return x.wt > y.wt; });
        std::vector<int> ranks(world);
        std::iota(ranks.begin(), ranks.end(), 0);
        std::stable_sort(ranks.begin(), ranks.end(), [&](int x, int y) { return load[x] < load[y]; });
        for (int k = 0; k < world; ++k) {
            const Seg &sg = segs[k];
            for (int B = sg.b0; B < sg.b1; ++B) owner[(size_t)A * (A + 1) / 2 + B] = ranks[k];
            load[ranks[k]] += sg.wt;
        }
    }
}

extern "C" {

int tf_version(void) { return 100; }

int tf_shard_plan_pairs(int n_shells, const int32_t *dim, int layout, int world, int32_t *owner)
{
    if (n_shells < 0 || world < 1 || !dim || !owner || layout < 0 || layout > 1) return TF_EINVAL;
    std::vector<int> d(dim, dim + n_shells), o;
    shard_plan_pairs(d, layout >= 1, world, o);
    std::copy(o.begin(), o.end(), owner);
    return TF_OK;
}

int tf_shard_plan(int n_blocks, const int64_t *weight, int world, int32_t *owner)
{
    if (n_blocks < 0 || world < 1 || !weight || !owner) return TF_EINVAL;
    std::vector<long long> w(weight, weight + n_blocks);
    std::vector<int> o;
    shard_plan(w, world, o);
    std::copy(o.begin(), o.end(), owner);
    return TF_OK;
}

tf_ctx *tf_create(int device, int rank, int world)
{
    // (GPU_MAX_HW_QUEUES -- more hardware queues for the concurrent launches of the tensor build -- belongs to the HOST application: it has
    // to be in the environment before the first HIP call of the process, and setenv from a library is not safe beside a threaded host's
    // getenv.  INTEGRATION.md recommends 16; the Python package sets it on import unless TUNA_NO_HWQ is set.)
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_create_error = std::string("no HIP device available (") + hipGetErrorString(e) + "); libtunafock has no CPU fallback";
        return nullptr;
    }
    if (device < 0 || device >= n || world < 1 || rank < 0 || rank >= world) {
        g_create_error = "tf_create: bad device ordinal or rank/world";
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        g_create_error = "hipSetDevice failed";
        return nullptr;
    }
    tf_ctx *ctx = new tf_ctx();
    ctx->device = device; ctx->rank = rank; ctx->world = world;
    ++g_live_contexts;
    return ctx;
}

void tf_destroy(tf_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    free_eri(ctx);
    if (ctx->d_eri) { (void)tf_free(ctx->d_eri); ctx->d_eri = nullptr; ctx->eri_cap = 0; }
    free_basis(ctx);
    tfscf::release(ctx->scf);
    for (auto &wsp : ctx->scf_batch) if (wsp) tfscf::release(*wsp);
    ctx->scf_batch.clear();
    tfdft::release(ctx->grid);
    for (hipEvent_t e : ctx->prof_ev) (void)hipEventDestroy(e);
    if (ctx->have_streams)
        for (int k = 0; k < tf_ctx::NSTREAM_MAX; ++k) { (void)hipStreamDestroy(ctx->streams[k]); (void)hipEventDestroy(ctx->sev[k]); }
    for (int k = 0; k < 3; ++k)
        if (ctx->scr[k]) (void)tf_free(ctx->scr[k]);
    if (ctx->mo_pool) (void)tf_free(ctx->mo_pool);
    if (ctx->comm) { (void)hipDeviceSynchronize(); (void)tfrccl::CommDestroy(ctx->comm); ctx->comm = nullptr; }
    if (ctx->d_jkstage) (void)tf_free(ctx->d_jkstage);
    if (ctx->d_agree) (void)tf_free(ctx->d_agree);
    if (ctx->d_lrec) (void)tf_free(ctx->d_lrec);
    if (ctx->d_tup) (void)tf_free(ctx->d_tup);
    if (ctx->d_gtab) (void)tf_free(ctx->d_gtab);
    ctx->arena1e.release();
    delete ctx;
    if (--g_live_contexts == 0) tfcache::trim();      // the last context of the process: the cached blocks go back to the runtime
}

const char *tf_last_error(const tf_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int tf_normalize(int l, int m, int n, int nprim, const double *exps, double *coefs_inout, double *norm_out)
{
    if (l < 0 || m < 0 || n < 0 || nprim <= 0 || !exps || !coefs_inout || !norm_out) return TF_EINVAL;
    tf::normalize_ao(l, m, n, nprim, exps, coefs_inout, norm_out);
    return TF_OK;
}

int tf_set_basis(tf_ctx *ctx, int n_ao_cart, const double *origin, const int32_t *lmn, const int32_t *prim_off,
                 const double *exps, const double *coefs_raw)
{
    if (!ctx) return TF_EINVAL;
    if (!origin || !lmn || !prim_off || !exps || !coefs_raw) TF_FAIL(ctx, TF_EINVAL, "tf_set_basis: null argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    free_eri(ctx);
    free_basis(ctx);
    tfdft::release(ctx->grid);
    std::string msg = tf::build_basis(ctx->bs, n_ao_cart, origin, lmn, prim_off, exps, coefs_raw);
    if (!msg.empty()) TF_FAIL(ctx, msg.find("aligned") != std::string::npos ? TF_EGEOM : TF_EINVAL, "%s", msg.c_str());
    const tf::Basis &bs = ctx->bs;
    std::vector<DShell> hs(bs.shells.size());
    for (size_t i = 0; i < hs.size(); ++i) hs[i] = {bs.shells[i].L, bs.shells[i].ncomp, bs.shells[i].comp_off, bs.shells[i].cart_off};
    std::vector<DPair> hp(bs.pairs.size());
    for (size_t i = 0; i < hp.size(); ++i) {
        const tf::Pair &p = bs.pairs[i];
        const tf::Shell &sa = bs.shells[p.A], &sb = bs.shells[p.B];
        hp[i] = {p.A, p.B, p.La, p.Lb, p.npp, p.pp_off, p.nE, 0, p.e_off, sa.ncomp, sb.ncomp, sa.comp_off, sb.comp_off, sa.cart_off, sb.cart_off, 0, 0};
    }
    // shell-pair classes: every pair of a class has the same angular momenta, component counts and contraction depth
    ctx->pair_class.assign(hp.size(), 0);
    ctx->class_pairs.clear();
    {
        // (shells that are not complete -- the one-component shells of the DECONTRACT ordering -- carry their components in the key: the
        // pairs of a class share one set of component-pair tables)
        auto comp_code = [&](const tf::Shell &sh) {
            if (sh.full) return 0;
            int code = 1;
            for (int c = 0; c < sh.ncomp && c < 2; ++c)
                code = code * 4096 + (bs.c_lx[sh.comp_off + c] | (bs.c_ly[sh.comp_off + c] << 4) | (bs.c_lz[sh.comp_off + c] << 8));
            return code;
        };
        std::map<std::array<int, 7>, int> ids;
        for (size_t i = 0; i < hp.size(); ++i) {
            const std::array<int, 7> key{hp[i].La, hp[i].Lb, getenv("TF_ERI_EXACT_CLASS") ? hp[i].npp : (hp[i].npp == 1 ? 1 : 0), hp[i].nca, hp[i].ncb,
                                         comp_code(bs.shells[bs.pairs[i].A]), comp_code(bs.shells[bs.pairs[i].B])};
            auto it = ids.find(key);
            if (it == ids.end()) { it = ids.emplace(key, (int)ids.size()).first; ctx->class_pairs.emplace_back(); }
            hp[i].cls = it->second;
            ctx->pair_class[i] = it->second;
            ctx->class_pairs[it->second].push_back((int)i);
        }
    }
    std::vector<double> boys;
    tf::boys_table(boys);
    DShell *d_sh; DPair *d_pr; int8_t *d_lx, *d_ly, *d_lz; double *d_sc, *d_p, *d_Pz, *d_K, *d_E, *d_boys;
    int rc;
    // per-L spherical rows as CSR over the Cartesian components of one shell
    std::vector<int> sph_base(TF_MAX_L + 2, 0), sph_ptr{0}, sph_idx;
    std::vector<double> sph_val;
    {
        std::vector<double> blk;
        for (int L = 0; L <= TF_MAX_L; ++L) {
            sph_base[L] = (int)sph_ptr.size() - 1;
            tf::sph_block(L, blk);
            const int nc = (L + 1) * (L + 2) / 2;
            for (int r = 0; r < 2 * L + 1; ++r) {
                for (int c = 0; c < nc; ++c)
                    if (blk[(size_t)r * nc + c] != 0.0) { sph_idx.push_back(c); sph_val.push_back(blk[(size_t)r * nc + c]); }
                sph_ptr.push_back((int)sph_idx.size());
            }
        }
    }
    // component-pair tables of every shell pair (eri_cfact_kernel): index word, normalisation ratio, position, parity-class order
    std::vector<int> ct_ix, ct_pos, ct_ord;
    std::vector<double> ct_sc;
    for (size_t i = 0; i < hp.size(); ++i) {
        DPair &pr = hp[i];
        const int nab = pr.nca * pr.ncb, L2 = pr.Lb + 1;
        pr.tab_off = (int)ct_ix.size();
        std::vector<int> cls(nab);
        for (int ca = 0; ca < pr.nca; ++ca)
            for (int cb = 0; cb < pr.ncb; ++cb) {
                const int a = pr.compoff_a + ca, b = pr.compoff_b + cb;
                const int ux = bs.c_lx[a], uy = bs.c_ly[a], uz = bs.c_lz[a], wx = bs.c_lx[b], wy = bs.c_ly[b], wz = bs.c_lz[b];
                const int c = ((ux + wx) & 1) | (((uy + wy) & 1) << 1);
                cls[ca * pr.ncb + cb] = c;
                ct_ix.push_back((ux * L2 + wx) | ((uy * L2 + wy) << 8) | ((uz * L2 + wz) << 16) | (c << 24));
                ct_pos.push_back((ca << 8) | cb);
                ct_sc.push_back(bs.c_scale[a] * bs.c_scale[b]);
            }
        pr.pcls[0] = 0;
        for (int c = 0; c < 4; ++c) {
            int n = 0;
            for (int f = 0; f < nab; ++f)
                if (cls[f] == c) { ct_ord.push_back(f); ++n; }
            pr.pcls[c + 1] = pr.pcls[c] + n;
        }
    }
    int *d_cti, *d_ctp, *d_cto; double *d_cts;
    if ((rc = upload(ctx, ct_ix, &d_cti)) || (rc = upload(ctx, ct_pos, &d_ctp)) || (rc = upload(ctx, ct_ord, &d_cto)) || (rc = upload(ctx, ct_sc, &d_cts)))
        return rc;
    int *d_sb, *d_sp, *d_si; double *d_sv;
    if ((rc = upload(ctx, hs, &d_sh)) || (rc = upload(ctx, hp, &d_pr)) || (rc = upload(ctx, bs.c_lx, &d_lx)) ||
        (rc = upload(ctx, bs.c_ly, &d_ly)) || (rc = upload(ctx, bs.c_lz, &d_lz)) || (rc = upload(ctx, bs.c_scale, &d_sc)) ||
        (rc = upload(ctx, bs.pp_p, &d_p)) || (rc = upload(ctx, bs.pp_Pz, &d_Pz)) || (rc = upload(ctx, bs.pp_K, &d_K)) ||
        (rc = upload(ctx, bs.epool, &d_E)) || (rc = upload(ctx, boys, &d_boys)) || (rc = upload(ctx, sph_base, &d_sb)) ||
        (rc = upload(ctx, sph_ptr, &d_sp)) || (rc = upload(ctx, sph_idx, &d_si)) || (rc = upload(ctx, sph_val, &d_sv)))
        return rc;
    ctx->db = DBasis{d_sh, d_pr, d_lx, d_ly, d_lz, d_sc, d_p, d_Pz, d_K, d_E, d_boys, d_sb, d_sp, d_si, d_sv, d_cti, d_ctp, d_cto, d_cts, nullptr, nullptr};
    ctx->host_pairs = hp;
    ctx->h_ct_ix = ct_ix; ctx->h_ct_ord = ct_ord; ctx->h_ct_sc = ct_sc;
    ctx->d_pairs = d_pr;
    ctx->have_basis = true;
    return TF_OK;
}

int tf_get_norms(const tf_ctx *ctx, double *norm, double *coefs_normalised)
{
    if (!ctx || !ctx->have_basis) return TF_EINVAL;
    if (norm) std::copy(ctx->bs.ao_norm.begin(), ctx->bs.ao_norm.end(), norm);
    if (coefs_normalised) std::copy(ctx->bs.ao_coef.begin(), ctx->bs.ao_coef.end(), coefs_normalised);
    return TF_OK;
}

int tf_dims(const tf_ctx *ctx, int *n_cart, int *n_sph, int *n_shell)
{
    if (!ctx || !ctx->have_basis) return TF_EINVAL;
    if (n_cart) *n_cart = ctx->bs.n_cart;
    if (n_sph) *n_sph = ctx->bs.n_sph;
    if (n_shell) *n_shell = (int)ctx->bs.shells.size();
    return TF_OK;
}

int tf_get_sph_matrix(const tf_ctx *ctx, double *U)
{
    if (!ctx || !ctx->have_basis || !U) return TF_EINVAL;
    std::vector<double> u;
    tf::dense_sph_matrix(ctx->bs, u);
    std::copy(u.begin(), u.end(), U);
    return TF_OK;
}

// AO-level CSR of U (or identity for Cartesian output) on the device
static int upload_csr(tf_ctx *ctx, int spherical)
{
    const tf::Basis &bs = ctx->bs;
    std::vector<int> ptr{0}, idx;
    std::vector<double> val;
    if (spherical) {
        std::vector<double> blk;
        for (const auto &sh : bs.shells) {
            tf::sph_block(sh.L, blk);
            for (int r = 0; r < sh.nsph; ++r) {
                for (int c = 0; c < sh.ncomp; ++c) {
                    const double v = blk[(size_t)r * sh.ncomp + c];
                    if (v != 0.0) { idx.push_back(sh.cart_off + c); val.push_back(v); }
                }
                ptr.push_back((int)idx.size());
            }
        }
    } else {
        for (int i = 0; i < bs.n_cart; ++i) { idx.push_back(i); val.push_back(1.0); ptr.push_back(i + 1); }
    }
    for (void *p : {(void *)ctx->d_csr_ptr, (void *)ctx->d_csr_idx, (void *)ctx->d_csr_val})
        if (p) (void)tf_free(p);
    int rc;
    if ((rc = upload(ctx, ptr, &ctx->d_csr_ptr, false)) || (rc = upload(ctx, idx, &ctx->d_csr_idx, false)) ||
        (rc = upload(ctx, val, &ctx->d_csr_val, false)))
        return rc;
    ctx->csr_rows = (int)ptr.size() - 1;
    return TF_OK;
}

static double seconds_between(hipEvent_t a, hipEvent_t b)
{
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms * 1e-3;
}

int tf_build_eri(tf_ctx *ctx, int spherical)
{
    if (!ctx) return TF_EINVAL;
    if (!ctx->have_basis) TF_FAIL(ctx, TF_EINVAL, "tf_build_eri: call tf_set_basis first");
    const tf::Basis &bs = ctx->bs;
    if (spherical && !bs.all_full)
        TF_FAIL(ctx, TF_EINVAL, "spherical output needs complete shells in canonical Cartesian order (use CARTHARM / spherical=0)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    free_eri(ctx);
    DBG("build_eri start");
    int rc = upload_csr(ctx, spherical);
    if (rc) return rc;
    DBG("csr uploaded");
    const int Nc = bs.n_cart, N = spherical ? bs.n_sph : bs.n_cart, ld = (N + 1) & ~1;
    if (ctx->grid.G > 0 && ctx->grid.N != N) tfdft::release(ctx->grid);   // (a grid of the other AO representation)
    ctx->spherical = spherical; ctx->N = N; ctx->ld = ld;
    const int npairs = (int)bs.pairs.size();

    // ---- which bra shell pairs (= row blocks) belong to this rank: longest-processing-time on row counts
    auto out_dim = [&](const tf::Shell &s) { return spherical ? s.nsph : s.ncomp; };
    auto out_off = [&](const tf::Shell &s) { return spherical ? s.sph_off : s.cart_off; };
    for (int p = 0; p < npairs; ++p) {
        ctx->host_pairs[p].outoff_a = out_off(bs.shells[bs.pairs[p].A]);
        ctx->host_pairs[p].outoff_b = out_off(bs.shells[bs.pairs[p].B]);
    }
    HIPCHK(ctx, hipMemcpy(ctx->d_pairs, ctx->host_pairs.data(), (size_t)npairs * sizeof(DPair), hipMemcpyHostToDevice));
    // layout: packed (8-fold unique, tf_jkpacked.hip.h) unless asked otherwise
    int layout = ctx->layout_req;
    if (const char *e = getenv("TF_ERI_LAYOUT")) layout = (e[0] == 't') ? 2 : ((e[0] == 'p') ? 1 : (e[0] == 'r' ? 0 : layout));
    if (layout < 0) layout = 1;                                       // (tiles: opt-in -- measured slower than packed at N = 400)
    ctx->layout = layout;
    // packed: the symmetry-unique, parity-allowed values (layouts 1 and 2: the generation and its slab are the same); tiles: stored
    // j-innermost in (i, class pair, strip, chunk) regions (tf_tiles.h) instead of row by row (tf_jkpacked.hip.h)
    const bool packed = layout >= 1, tiles = layout == 2;
    std::vector<long long> pair_rows(npairs), pair_weight(npairs);
    for (int p = 0; p < npairs; ++p) {
        const tf::Shell &a = bs.shells[bs.pairs[p].A], &b = bs.shells[bs.pairs[p].B];
        pair_rows[p] = (bs.pairs[p].A == bs.pairs[p].B) ? (long long)out_dim(a) * (out_dim(a) + 1) / 2
                                                        : (long long)out_dim(a) * out_dim(b);
        long long w = 0;                                         // stored elements of the block's rows
        for (int x = 0; x < out_dim(a); ++x)
            for (int y = 0; y < out_dim(b); ++y) {
                const long long i = out_off(a) + x, j = out_off(b) + y;
                if (i >= j) w += packed ? packed_row_len(i, j) : 1;
            }
        pair_weight[p] = w;
    }
    std::vector<int> owner;
    {
        std::vector<int> dims(bs.shells.size());
        for (size_t q = 0; q < dims.size(); ++q) dims[q] = out_dim(bs.shells[q]);
        shard_plan_pairs(dims, packed, ctx->world, owner);           // pair index A(A+1)/2 + B = position in bs.pairs
    }
    ctx->my_pairs.clear();
    for (int p = 0; p < npairs; ++p)
        if (owner[p] == ctx->rank) ctx->my_pairs.push_back(p);

    // ---- parity classes of the output AOs and the layout tables (packed layout)
    if (packed) {
        std::vector<int> cls(N);
        // every Cartesian component of a real spherical AO has the AO's x/y parity: the first one decides
        {
            std::vector<double> blk;
            int o = 0;
            for (const auto &sh : bs.shells) {
                const int nout = out_dim(sh);
                if (spherical) tf::sph_block(sh.L, blk);
                for (int r = 0; r < nout; ++r) {
                    int cc = r;
                    if (spherical) {
                        cc = 0;
                        while (cc < sh.ncomp && blk[(size_t)r * sh.ncomp + cc] == 0.0) ++cc;
                    }
                    const int ca = sh.cart_off + cc;
                    cls[o++] = (bs.ao_lmn[3 * ca] & 1) | ((bs.ao_lmn[3 * ca + 1] & 1) << 1);
                }
            }
        }
        if ((rc = build_blocked_layout(ctx, cls))) return rc;
    }
    const tf_ctx::HostLayout &H = ctx->hl;
    // ---- row tables
    std::vector<int2> row_ij;                                       // original (i >= j) of every local row
    std::vector<int> rowmap((size_t)N * (N + 1) / 2, -1);
    std::vector<long long> pair_first_row(npairs, -1);
    for (int p : ctx->my_pairs) {
        const tf::Shell &a = bs.shells[bs.pairs[p].A], &b = bs.shells[bs.pairs[p].B];
        pair_first_row[p] = (long long)row_ij.size();
        for (int x = 0; x < out_dim(a); ++x)
            for (int y = 0; y < out_dim(b); ++y) {
                const int i = out_off(a) + x, j = out_off(b) + y;
                if (i < j) continue;
                rowmap[(size_t)i * (i + 1) / 2 + j] = (int)row_ij.size();
                row_ij.push_back(make_int2(i, j));
            }
    }
    std::vector<long long> rowoff;
    std::vector<int> rowsec, rowlen;
    auto ikey = [](int x, int y) { const int hi = std::max(x, y), lo = std::min(x, y); return (size_t)hi * (hi + 1) / 2 + lo; };
    if (packed) {
        // owned rows in ascending internal (sigma(i), sigma(j)): rows that share i and the class of j are adjacent (the row groups of the
        // J/K kernel); rowmap is keyed by the unordered pair of internal indices
        {
            // (the keys sigma(i) N + sigma(j) are distinct: one pass over the N^2 key space instead of a comparison sort -- 3 ms at N = 400)
            std::vector<int> slot((size_t)N * N, -1);
            for (size_t r = 0; r < row_ij.size(); ++r) slot[(size_t)H.sigma[row_ij[r].x] * N + H.sigma[row_ij[r].y]] = (int)r;
            std::vector<int2> sorted;
            sorted.reserve(row_ij.size());
            for (size_t k = 0; k < slot.size(); ++k)
                if (slot[k] >= 0) sorted.push_back(row_ij[(size_t)slot[k]]);
            row_ij.swap(sorted);
        }
        std::fill(rowmap.begin(), rowmap.end(), -1);
        rowoff.assign(row_ij.size() + 1, 0);
        rowsec.assign(6 * row_ij.size() + 6, 0);
        rowlen.assign(row_ij.size() + 1, 0);
        for (size_t r = 0; r < row_ij.size(); ++r) {
            const int i = row_ij[r].x, j = row_ij[r].y;
            rowmap[ikey(H.sigma[i], H.sigma[j])] = (int)r;
            if (!tiles) rowlen[r] = H.row_shape(H.cls[i] ^ H.cls[j], H.sigma[i], &rowsec[6 * r]);
        }
    }
    if (tiles) {
        // regions of the stored tensor and the task list of the Fock kernel (tf_tiles_host.h)
        tft::ClassInfo C;
        C.N = N;
        for (int q = 0; q < 4; ++q) { C.cstart[q] = H.cstart[q]; C.csize[q] = H.csize[q]; }
        C.clsI = H.clsI; C.origI = H.origI; C.cntA = H.cntA;
        std::vector<std::pair<int, int>> rows_ij(row_ij.size());
        for (size_t r = 0; r < row_ij.size(); ++r) rows_ij[r] = {H.sigma[row_ij[r].x], H.sigma[row_ij[r].y]};
        static const int part_steps = std::min(TT_STEPS_MAX, getenv("TF_TILE_PART_STEPS") ? std::max(1, atoi(getenv("TF_TILE_PART_STEPS"))) : TT_STEPS_MAX);
        ctx->tclass = C; ctx->trows = rows_ij;
        std::string e = tft::build(C, rows_ij, part_steps, ctx->tiles);
        if (!e.empty()) TF_FAIL(ctx, TF_EINVAL, "%s", e.c_str());
        static const int ksub_env = getenv("TF_TILE_KSUB") ? atoi(getenv("TF_TILE_KSUB")) : 32;     // (measured at N = 400: strips of 32 rows 2.2 ms, of 64 3.0, of 16 2.3)
        ctx->ksub1 = (ksub_env == 64 || ksub_env == 16) ? ksub_env : 32;
        if (ctx->ksub1 != TT_KS) {
            e = tft::build_list(C, ctx->tiles, rows_ij, ctx->ksub1, part_steps, ctx->tsub1);
            if (!e.empty()) TF_FAIL(ctx, TF_EINVAL, "%s", e.c_str());
        }
        if (ctx->tiles.max_slice * (long long)part_steps * 8 > 0x7fffffffLL) TF_FAIL(ctx, TF_EINVAL, "tiles layout: a task's region exceeds the 32-bit offsets of the Fock kernel");
        ctx->n_elems = ctx->tiles.n_elems;
        const tft::Tables &TT = ctx->tiles;
        std::vector<int> tab(TVT_LEN, 0);
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) tab[TVT_PID + 4 * a + b] = std::max(0, TT.pid[a][b]);
        for (int p2 = 0; p2 < TT.npair; ++p2) { tab[TVT_PA + p2] = TT.pa[p2]; tab[TVT_PB + p2] = TT.pb[p2]; tab[TVT_PMOFF + p2] = TT.pm_off[p2]; tab[TVT_PMPITCH + p2] = TT.pm_pitch[p2]; }
        for (int q = 0; q < 4; ++q) { tab[TVT_CSTART + q] = H.cstart[q]; tab[TVT_CSIZE + q] = H.csize[q]; }
        auto up_t = [&](const auto &h, auto **d) -> int {
            int rc2 = upload(ctx, h, d, false);
            if (!rc2) ctx->tile_allocs.push_back(*d);
            return rc2;
        };
        TTask *d_regions = nullptr; TPairI *d_pp = nullptr; TRunI *d_rr = nullptr;
        if ((rc = up_t(TT.primary.tasks_by_region, &d_regions)) || (rc = up_t(TT.primary.pairs, &d_pp)) || (rc = up_t(TT.primary.runs, &d_rr)) ||
            (rc = up_t(tab, &ctx->d_tvtab)))
            return rc;
        TView V{};
        V.regions = d_regions; V.prim_pairs = d_pp; V.prim_runs = d_rr; V.edge_base = TT.edge_base; V.N = N; V.pm_len = TT.pm_len; V.tab = ctx->d_tvtab;
        ctx->tv = V;
        ctx->tl1.d_pairs = d_pp; ctx->tl1.d_runs = d_rr;
    }
    if (packed && !tiles) {
        // storage units: runs of up to 8 consecutive j of one class with the same i, cut from the top (the row groups of the kernel; its
        // groups of 4 for two densities are halves of them); the rows of a unit are interleaved segment by segment
        long long off = 0;
        for (long long r = (long long)row_ij.size() - 1; r >= 0;) {
            long long r0 = r;
            auto sI = [&](long long q) { return H.sigma[row_ij[q].x]; };
            auto sJ = [&](long long q) { return H.sigma[row_ij[q].y]; };
            while (r0 > 0 && sI(r0 - 1) == sI(r) && sJ(r0 - 1) == sJ(r0) - 1 && H.clsI[sJ(r0 - 1)] == H.clsI[sJ(r)] && r - r0 + 1 < TF_JKP_JBB) --r0;
            const int nr = (int)(r - r0 + 1);
            for (long long q = r0; q <= r; ++q) { rowoff[q] = off; rowsec[6 * q + 4] = (int)(q - r0); rowsec[6 * q + 5] = nr; }
            off += (long long)nr * rowlen[r0];
            r = r0 - 1;
        }
        rowoff[row_ij.size()] = off;
        ctx->n_elems = off;
    }
    // work tables of the J/K kernel for groups of RB rows (8: one density per pass; 4: two)
    auto build_jk_tables = [&](int RB, tf_ctx::JKTables &T) -> int {
        std::vector<JKGroup> groups;
        std::vector<JKTask> tasks;
        std::vector<JKSuper> supers;
        std::vector<int> gfirst(2 * (size_t)N, 0);
        long long ypart_len = 0;
        auto sI = [&](long long r) { return H.sigma[row_ij[r].x]; };
        auto sJ = [&](long long r) { return H.sigma[row_ij[r].y]; };
        // groups: runs of consecutive internal j of one class with the same i, largest j first
        for (long long r = (long long)row_ij.size() - 1; r >= 0;) {
            long long r0 = r;
            while (r0 > 0 && sI(r0 - 1) == sI(r) && sJ(r0 - 1) == sJ(r0) - 1 && H.clsI[sJ(r0 - 1)] == H.clsI[sJ(r)] && r - r0 + 1 < RB) --r0;
            JKGroup g{};
            g.i = sI(r); g.j0 = sJ(r0); g.nr = (int)(r - r0 + 1); g.r0 = (int)r0;
            g.c = H.clsI[g.i] ^ H.clsI[g.j0]; g.lamj0 = g.j0 - H.cstart[H.clsI[g.j0]];
            g.ub = rowoff[r0]; g.p0 = rowsec[6 * (size_t)r0 + 4]; g.unr = rowsec[6 * (size_t)r0 + 5];
            for (int a = 0; a < 4; ++a) g.secoff[a] = rowsec[6 * (size_t)r0 + a];
            groups.push_back(g);
            r = r0 - 1;
        }
        // (the reductions want the groups of one i contiguous: they are, the rows being sorted by i)
        for (size_t gi = 0; gi < groups.size(); ++gi) {
            const int a = groups[gi].i;
            if (gfirst[N + a] == gfirst[a]) gfirst[a] = (int)gi;
            gfirst[N + a] = (int)gi + 1;
        }
        // super-groups: up to TF_JKP_GPW * TF_JKP_W adjacent groups (TF_JKP_GPW per wave) with the same i and class share a workgroup and one Jt partial.  The kernel's
        // groups index into `groups`, so the super list may be reordered freely: by class, then by descending original i (the Jt
        // reduction needs those that reach an AO k to be a prefix of their class's list)
        for (size_t gi = 0; gi < groups.size();) {
            size_t ge = gi + 1;
            while (ge < groups.size() && groups[ge].i == groups[gi].i && groups[ge].c == groups[gi].c && ge - gi < TF_JKP_GPW * TF_JKP_W) ++ge;
            JKSuper sg{};
            sg.g0 = (int)gi; sg.ng = (int)(ge - gi); sg.c = groups[gi].c; sg.i = groups[gi].i;
            for (int a = 0; a < 4; ++a) sg.ke[a] = H.ke(a, sg.i);
            supers.push_back(sg);
            gi = ge;
        }
        std::stable_sort(supers.begin(), supers.end(), [&](const JKSuper &u, const JKSuper &v) {
            return u.c != v.c ? u.c < v.c : H.origI[u.i] > H.origI[v.i];
        });
        JKJtPlan jp{};
        for (size_t si = 0; si < supers.size(); ++si) {
            supers[si].yoff = ypart_len;
            ypart_len += H.NP[supers[si].c];
            ++jp.sfirst[supers[si].c + 1];
        }
        for (int c = 0; c < 4; ++c) {
            jp.sfirst[c + 1] += jp.sfirst[c];
            jp.bfirst[c + 1] = jp.bfirst[c] + (int)((H.NP[c] + TF_JKR_THREADS - 1) / TF_JKR_THREADS);
        }
        // tasks (super-group, chunk) that have at least one step, longest first: the hardware dispatches workgroups in this order
        std::vector<int> steps;
        for (size_t si = 0; si < supers.size(); ++si)
            for (int w = 0; w < H.NW; ++w)
                if (H.task_exists(supers[si].c, w, supers[si].i)) {
                    const int walk = H.ke(H.chunk_cls[w] ^ supers[si].c, supers[si].i) - H.kap0[(size_t)supers[si].c * H.NW + w];
                    for (int part = 0; part * H.KS < walk; ++part) {
                        tasks.push_back(JKTask{(int)si, w, part, 0});
                        steps.push_back(std::min(H.KS, walk - part * H.KS));
                    }
                }
        {
            // workgroups of 4, 2 or 1 waves (two groups per wave): a rank of several holds few groups per (i, class), and a wave without
            // a group would only sit in the barriers of its workgroup and occupy a SIMD slot.  One launch per workgroup size; inside
            // a launch longest first.
            auto waves = [&](int t) {                                  // workgroup sizes TF_JKP_W, TF_JKP_W / 2, TF_JKP_W / 4 waves (at least one)
                const int nwv = (supers[tasks[t].super].ng + TF_JKP_GPW - 1) / TF_JKP_GPW;
                for (int b = 2; b >= 0; --b) if ((TF_JKP_W >> b) >= 1 && nwv <= (TF_JKP_W >> b)) return TF_JKP_W >> b;
                return TF_JKP_W;
            };
            std::vector<int> ord(tasks.size());
            std::iota(ord.begin(), ord.end(), 0);
            std::stable_sort(ord.begin(), ord.end(), [&](int u, int v) { return waves(u) != waves(v) ? waves(u) > waves(v) : steps[u] > steps[v]; });
            std::vector<JKTask> sorted(tasks.size());
            T.bucket[0] = 0; T.bucket[1] = T.bucket[2] = T.bucket[3] = (int)tasks.size();
            for (size_t t = 0; t < ord.size(); ++t) {
                sorted[t] = tasks[ord[t]];
                const int wv = waves(ord[t]);
                if (wv <= TF_JKP_W / 2 && T.bucket[1] == (int)tasks.size()) T.bucket[1] = (int)t;
                if (wv <= TF_JKP_W / 4 && T.bucket[2] == (int)tasks.size()) T.bucket[2] = (int)t;
            }
            if (T.bucket[2] < T.bucket[1]) T.bucket[1] = T.bucket[2];
            tasks.swap(sorted);
        }
        // the class-diagonal list: the same tasks in the same order without those whose column class is neither i's nor j's
        std::vector<JKTask> tasks_cd;
        tasks_cd.reserve(tasks.size());
        for (int b = 0; b < 4; ++b) T.bucket_cd[b] = 0;
        for (size_t t = 0; t < tasks.size(); ++t) {
            for (int b = 1; b < 4; ++b) if ((int)t == T.bucket[b]) T.bucket_cd[b] = (int)tasks_cd.size();
            const JKSuper &sg = supers[tasks[t].super];
            const int ci = H.clsI[sg.i], cj = ci ^ sg.c, cb = H.chunk_cls[tasks[t].w];
            if (sg.c == 0 || cb == ci || cb == cj) tasks_cd.push_back(tasks[t]);
        }
        for (int b = 1; b < 4; ++b) if (T.bucket[b] == (int)tasks.size()) T.bucket_cd[b] = (int)tasks_cd.size();
        T.n_tasks_cd = (int)tasks_cd.size();
        if (g_dbg) {
            long long st_all = 0, st_cd = 0;                       // wave steps of the two lists (what a pass costs)
            for (size_t t = 0; t < tasks.size(); ++t) {
                const JKSuper &sg = supers[tasks[t].super];
                const int ci = H.clsI[sg.i], cj = ci ^ sg.c, cb = H.chunk_cls[tasks[t].w];
                const int walk = H.ke(cb ^ sg.c, sg.i) - H.kap0[(size_t)sg.c * H.NW + tasks[t].w];
                const long long stp = (long long)std::min(H.KS, walk - tasks[t].part * H.KS) * ((sg.ng + TF_JKP_GPW - 1) / TF_JKP_GPW);
                st_all += stp;
                if (sg.c == 0 || cb == ci || cb == cj) st_cd += stp;
            }
            DBG("J/K tasks: %zu (class-diagonal list: %zu), wave steps %lld (%lld)", tasks.size(), tasks_cd.size(), st_all, st_cd);
        }
        int rc2;
        if ((rc2 = upload(ctx, groups, &T.d_groups, false)) || (rc2 = upload(ctx, gfirst, &T.d_gfirst, false)) ||
            (rc2 = upload(ctx, tasks, &T.d_tasks, false)) || (rc2 = upload(ctx, supers, &T.d_supers, false)) ||
            (rc2 = upload(ctx, tasks_cd, &T.d_tasks_cd, false)))
            return rc2;
        T.n_groups = (int)groups.size(); T.n_tasks = (int)tasks.size(); T.n_supers = (int)supers.size();
        T.nseg = std::max(1, std::min(TF_JKP_SEG, T.n_supers / 128));
        T.ypart_len = ypart_len;
        T.jp = jp;
        return TF_OK;
    };
    ctx->n_rows = (long long)row_ij.size();
    const long long row_len = (long long)N * ld;
    if (!packed) ctx->n_elems = ctx->n_rows * row_len;
    if (ctx->n_rows > 0x7fffffffLL) TF_FAIL(ctx, TF_EINVAL, "too many tensor rows for this build");
    {
        // the tensor buffer is kept across builds (geometry scans rebuild a tensor of the same size); it is replaced when it is
        // too small or more than twice too large
#ifdef TF_ABL_ALLVALID
        const size_t need = std::max<size_t>(1, (size_t)ctx->n_elems * sizeof(double)) + (4u << 20);
#else
        const size_t need = std::max<size_t>(1, (size_t)ctx->n_elems * sizeof(double));
#endif
        if (ctx->d_eri && (ctx->eri_cap < need || ctx->eri_cap > 2 * need + (64u << 20))) {
            (void)tf_free(ctx->d_eri);
            ctx->d_eri = nullptr; ctx->eri_cap = 0;
        }
        if (!ctx->d_eri) {
            HIPCHK(ctx, tf_malloc((void **)&ctx->d_eri, need));
            ctx->eri_cap = need;
        }
    }
    if ((rc = upload(ctx, row_ij, &ctx->d_row_ij, false)) || (rc = upload(ctx, rowmap, &ctx->d_rowmap, false))) return rc;
    if (packed) {
        if (!tiles && ((rc = upload(ctx, rowoff, &ctx->d_rowoff, false)) || (rc = upload(ctx, rowsec, &ctx->d_rowsec, false)))) return rc;
        ctx->db.bl = ctx->bl;
        ctx->db.RLS = H.RLS;
    }
    if (tiles) HIPCHK(ctx, hipMemsetAsync(ctx->d_eri, 0, (size_t)ctx->n_elems * sizeof(double), 0));   // the pad slots of the rows and pieces stay zero
    // The tables only the CONSUMERS of the tensor need (work tables of the J/K kernel, reduction lists, rows by class) are built on the
    // host while the generation kernels run: called behind the launches of the last slab (7 ms of host time at N = 400).
    auto consumer_tables = [&]() -> int {
        if (!packed) return TF_OK;
        int rc2;
        {
            // rows listed class by class (the AO->MO transformation works on one class at a time: a row of class c is nonzero only in
            // the blocks (k of class a) x (l of class a ^ c))
            std::vector<int> class_rows, row_pos(row_ij.size(), 0);
            class_rows.reserve(row_ij.size());
            for (int c = 0; c < 4; ++c) {
                ctx->class_row_off[c] = (long long)class_rows.size();
                for (size_t r = 0; r < row_ij.size(); ++r)
                    if ((H.cls[row_ij[r].x] ^ H.cls[row_ij[r].y]) == c) { row_pos[r] = (int)class_rows.size(); class_rows.push_back((int)r); }
            }
            ctx->class_row_off[4] = (long long)class_rows.size();
            if ((rc2 = upload(ctx, class_rows, &ctx->d_class_rows, false)) || (rc2 = upload(ctx, row_pos, &ctx->d_row_pos, false))) return rc2;
        }
        if (tiles) {
            const tft::Tables &TT = ctx->tiles;
            const tft::TaskList &TL = ctx->ksub1 == TT_KS ? TT.primary : ctx->tsub1;
            auto up_t = [&](const auto &h, auto **d) -> int {
                int rc3 = upload(ctx, h, d, false);
                if (!rc3) ctx->tile_allocs.push_back(*d);
                return rc3;
            };
            if ((rc2 = tile_list_upload(ctx, TL, ctx->ksub1 != TT_KS, ctx->tl1)) || (rc2 = up_t(TT.jlist_ptr, &ctx->d_jlist_ptr)) || (rc2 = up_t(TT.jlist, &ctx->d_jlist)))
                return rc2;
            return TF_OK;
        }
        if ((rc2 = build_jk_tables(JKShape<1>::RB, ctx->jkt[0])) || (rc2 = build_jk_tables(JKShape<2>::RB, ctx->jkt[1]))) return rc2;
        // reduction table: the rows (z, x), z != x, listed by their second index x (internal)
        std::vector<int> jptr((size_t)N + 1, 0);
        std::vector<int2> jrows;
        for (size_t r = 0; r < row_ij.size(); ++r) {
            const int iI = H.sigma[row_ij[r].x], jI = H.sigma[row_ij[r].y];
            if (iI != jI) ++jptr[jI + 1];
        }
        for (int x = 0; x < N; ++x) jptr[x + 1] += jptr[x];
        jrows.assign((size_t)std::max(1, jptr[N]), make_int2(0, 0));
        {
            std::vector<int> fill(jptr.begin(), jptr.end() - 1);
            for (size_t r = 0; r < row_ij.size(); ++r) {            // (ascending local row: a fixed summation order)
                const int iI = H.sigma[row_ij[r].x], jI = H.sigma[row_ij[r].y];
                if (iI != jI) jrows[fill[jI]++] = make_int2((int)r, row_ij[r].x);
            }
        }
        if ((rc2 = upload(ctx, jptr, &ctx->d_jptr, false)) || (rc2 = upload(ctx, jrows, &ctx->d_jrows, false))) return rc2;
        {
            // dispatch order of the exchange reduction: the output rows with the most partial vectors (rows listed under x + groups of x) first
            std::vector<long long> work((size_t)N, 0);
            for (int x = 0; x < N; ++x) work[x] = jptr[x + 1] - jptr[x];
            for (const int2 &ij : row_ij) work[H.sigma[ij.x]] += 1;          // (8 rows of a group: weight 1/8 each would do; the order is what counts)
            std::vector<int> xorder((size_t)N);
            std::iota(xorder.begin(), xorder.end(), 0);
            std::stable_sort(xorder.begin(), xorder.end(), [&](int a, int b) { return work[a] > work[b]; });
            if ((rc2 = upload(ctx, xorder, &ctx->d_xorder, false))) return rc2;
        }
        return TF_OK;
    };

    DBG("rows=%lld N=%d ld=%d (tensor + row tables allocated)", ctx->n_rows, N, ld);
    // ---- slabs of bra pairs: Cartesian block -> ket transform -> bra transform -> tensor rows
    // Large problems launch per (bra class, ket class) with the ket transform fused into the ERI kernels: no Cartesian slab at all.
    bool per_class = (long long)ctx->my_pairs.size() * npairs >= 2000000LL;
    if (const char *m = getenv("TF_ERI_MODE")) per_class = (m[0] == 'c');
    // A slab row is one Cartesian bra component pair: Nc^2 Cartesian ket values (small-problem mode only) and the ket-transformed
    // row -- N x ld (rows layout) or the complete-row shape of the packed layout (RLS doubles, ~ N^2 / 8).  1 GiB of slab, more (up
    // to 4 GiB) for big tensors: fewer, larger class launches.
    const size_t cart_row_bytes = (size_t)Nc * Nc * sizeof(double);
    const size_t t2_row_bytes = (packed ? (size_t)H.RLS : (size_t)N * ld) * sizeof(double);
    const size_t slab_row_bytes = std::max<size_t>(8, std::max(t2_row_bytes, per_class ? (size_t)0 : cart_row_bytes));
    size_t slab_total = 0;                                       // this rank's bra rows: one slab if that is <= 4 GiB
    for (int p : ctx->my_pairs) slab_total += (size_t)bs.shells[bs.pairs[p].A].ncomp * bs.shells[bs.pairs[p].B].ncomp * slab_row_bytes;
    size_t slab_bytes = std::min<size_t>((size_t)4 << 30, std::max<size_t>((size_t)1 << 30, std::max((size_t)ctx->n_elems, slab_total)));
    if (const char *e = getenv("TF_SLAB_MB")) slab_bytes = (size_t)std::max(1, atoi(e)) << 20;
    long long max_rows_c = std::max<long long>(1, (long long)(slab_bytes / slab_row_bytes));
    long long biggest = 1;
    for (int p : ctx->my_pairs)
        biggest = std::max<long long>(biggest, (long long)bs.shells[bs.pairs[p].A].ncomp * bs.shells[bs.pairs[p].B].ncomp);
    max_rows_c = std::max(max_rows_c, biggest);
    {
        long long need = 0;                                      // never more than this rank's rows need
        for (int p : ctx->my_pairs) need += (long long)bs.shells[bs.pairs[p].A].ncomp * bs.shells[bs.pairs[p].B].ncomp;
        max_rows_c = std::max<long long>(biggest, std::min(max_rows_c, need));
    }
    if ((rc = ensure_scratch(ctx, 0, per_class ? 8 : (size_t)max_rows_c * cart_row_bytes)) ||
        (rc = ensure_scratch(ctx, 2, (size_t)max_rows_c * t2_row_bytes)))
        return rc;
    double *d_C = ctx->scr[0], *d_T2 = ctx->scr[2];
    double t_stage[4] = {0, 0, 0, 0};
    long long n_quart = 0, n_primq = 0, n_compq = 0;
    // work counters: ket pairs / primitive pairs / component pairs with first shell <= A (the pair list is A-major).  The packed
    // layout computes exactly the kets whose first shell does not exceed the bra's; the rows layout all of them.
    const int nsh = (int)bs.shells.size();
    std::vector<long long> cum_pairs(nsh + 1, 0), cum_pp(nsh + 1, 0), cum_comp(nsh + 1, 0);
    for (int p = 0; p < npairs; ++p) {
        const int A = bs.pairs[p].A;
        cum_pairs[A + 1] += 1;
        cum_pp[A + 1] += bs.pairs[p].npp;
        cum_comp[A + 1] += (long long)bs.shells[A].ncomp * bs.shells[bs.pairs[p].B].ncomp;
    }
    for (int a = 0; a < nsh; ++a) { cum_pairs[a + 1] += cum_pairs[a]; cum_pp[a + 1] += cum_pp[a]; cum_comp[a + 1] += cum_comp[a]; }
    // Nominal operation count of the reference algorithm for the quartets this build evaluates: per primitive
    // AO quartet that passes the parity test (pyx:1324-1327), 8 x the inner terms of the loop nest pyx:1179-1217 -- (lx12/2+1)(lx34/2+1)
    // (ly12/2+1)(ly34/2+1)(lz12+1)(lz34+1), a product of a bra and a ket factor -- plus the Boys / R table cost 6 (L+1) + 3 (L+1)^2 / 2 + 60.
    // Per shell pair and parity class c: W = sum of its factor over the component pairs of the class, n = their number.
    std::vector<std::array<double, 4>> pairW(npairs), pairNc(npairs);
    for (int p = 0; p < npairs; ++p) {
        const tf::Shell &sa = bs.shells[bs.pairs[p].A], &sb = bs.shells[bs.pairs[p].B];
        pairW[p] = {0, 0, 0, 0}; pairNc[p] = {0, 0, 0, 0};
        for (int ca = 0; ca < sa.ncomp; ++ca)
            for (int cb = 0; cb < sb.ncomp; ++cb) {
                const int u = sa.comp_off + ca, v = sb.comp_off + cb;
                const int lx = bs.c_lx[u] + bs.c_lx[v], ly = bs.c_ly[u] + bs.c_ly[v], lz = bs.c_lz[u] + bs.c_lz[v];
                const int c = (lx & 1) | ((ly & 1) << 1);
                pairW[p][c] += (double)((lx / 2 + 1) * (ly / 2 + 1) * (lz + 1));
                pairNc[p][c] += 1.0;
            }
    }
    // prefix sums over the first shell A of the ket pairs (the pair list is A-major): npp x W per class, and npp x n per class and Lc + Ld
    std::vector<std::array<double, 4>> cumW(nsh + 1, std::array<double, 4>{0, 0, 0, 0});
    std::vector<std::array<double, 44>> cumNL(nsh + 1);
    for (auto &x : cumNL) x.fill(0.0);
    for (int p = 0; p < npairs; ++p) {
        const int A = bs.pairs[p].A, lcd = std::min(10, bs.pairs[p].La + bs.pairs[p].Lb);
        for (int c = 0; c < 4; ++c) {
            cumW[A + 1][c] += (double)bs.pairs[p].npp * pairW[p][c];
            cumNL[A + 1][4 * lcd + c] += (double)bs.pairs[p].npp * pairNc[p][c];
        }
    }
    for (int a = 0; a < nsh; ++a) {
        for (int c = 0; c < 4; ++c) cumW[a + 1][c] += cumW[a][c];
        for (int x = 0; x < 44; ++x) cumNL[a + 1][x] += cumNL[a][x];
    }
    double nominal_flops = 0.0;
    DBG("slab buffers allocated");
    auto t_wall0 = std::chrono::steady_clock::now();
    // class-sorted ket lists on the device (one contiguous range per class)
    const int ncls = (int)ctx->class_pairs.size();
    std::vector<int> ket_sorted, ket_off(ncls + 1, 0), cls_maxnpp(ncls, 1);
    for (int c = 0; c < ncls; ++c) {
        ket_sorted.insert(ket_sorted.end(), ctx->class_pairs[c].begin(), ctx->class_pairs[c].end());
        ket_off[c + 1] = (int)ket_sorted.size();
        for (int p : ctx->class_pairs[c]) cls_maxnpp[c] = std::max(cls_maxnpp[c], bs.pairs[p].npp);
    }
    auto pair_cost = [&](int p) {                                  // primitive pairs x components: what a quartet with this pair costs
        return (long long)bs.pairs[p].npp * bs.shells[bs.pairs[p].A].ncomp * bs.shells[bs.pairs[p].B].ncomp;
    };
    // groups of shell pairs: by La + Lb (0-1, 2-3, 4-5, 6-7, 8, 9, 10: the table sizes grow with the fourth power of L, and only the
    // very top -- (hh|hh), (hh|gh), ... -- exceeds LDS and falls back to the component-per-lane kernel) and by contracted / uncontracted
    constexpr int NLPG = 7, NGRP = 2 * NLPG;
    auto lp_group = [](int lp) { return lp <= 1 ? 0 : (lp <= 3 ? 1 : (lp <= 5 ? 2 : (lp <= 7 ? 3 : std::min(lp - 4, 6)))); };
    auto pair_group = [&](int p) { return 2 * lp_group(bs.pairs[p].La + bs.pairs[p].Lb) + (bs.pairs[p].npp > 1 ? 1 : 0); };
    int kets_goff[NGRP + 1] = {};
    int *d_kets = nullptr, *d_kets_all = nullptr;
    std::vector<int> kets_all_host;                              // same order as d_kets_all
    if ((rc = upload(ctx, ket_sorted, &d_kets, false))) return rc;
    {
        // generic (single-launch) mode: heaviest ket pairs first -- workgroups are dispatched in index order, and a deeply contracted
        // (pp|pp) quartet of Ar2/cc-pVQZ runs for 12 ms: it has to start early, not at the tail of the launch
        // the launches are made per (bra group, ket group) of shell pairs -- groups by La + Lb -- so that the LDS carve-out of each
        // launch fits its own angular momenta (the (gg|gg) tables need 100 KB, the (ss|ss) ones nothing)
        std::vector<int> all(npairs);
        std::iota(all.begin(), all.end(), 0);
        std::stable_sort(all.begin(), all.end(), [&](int x, int y) {
            const int gx = pair_group(x), gy = pair_group(y);
            return gx != gy ? gx < gy : pair_cost(x) > pair_cost(y);
        });
        for (int p : all) ++kets_goff[pair_group(p) + 1];
        for (int g = 0; g < NGRP; ++g) kets_goff[g + 1] += kets_goff[g];
        if ((rc = upload(ctx, all, &d_kets_all, false))) return rc;
        kets_all_host = all;
    }
    // Families of ket pairs (general contractions, eri_cfact_kernel<.., MM>): shells with identical primitives -- same centre, L, exponents,
    // components -- get the same primitive-set id; the contracted pairs of a group with the same (id, id) differ only in their primitive-pair
    // weights and in the AOs they write.  Per contracted group: heads (first member, the group's cost order kept), members of each head.
    constexpr int FAM_MM = 9, FAM_MA_CC = 3;
    // (on when the build has the work to fill the chip with fewer, longer workgroups: Ar2/cc-pVQZ -- 5.0e7 primitive shell quartets -- gains
    // 20 %, N2/cc-pVTZ -- 1.2e6 -- loses 20 %; TF_ERI_FAMILIES=0 / 1 forces)
    const double prim_quartets_est = 0.5 * (double)cum_pp[nsh] * (double)cum_pp[nsh];
    const bool fam_off = getenv("TF_ERI_FAMILIES") ? getenv("TF_ERI_FAMILIES")[0] == '0' : prim_quartets_est < 8.0e6;
    const bool cc_fam_off = getenv("TF_ERI_CC_FAMILIES") && getenv("TF_ERI_CC_FAMILIES")[0] == '0';
    const bool bra_fam_on = !fam_off && getenv("TF_ERI_BRA_FAMILIES") && getenv("TF_ERI_BRA_FAMILIES")[0] == '1';
    std::vector<int> fam_heads, fam_ptr{0}, fam_mem;
    int fam_goff[NGRP + 1] = {};
    int *d_fam_heads = nullptr, *d_fam_ptr = nullptr, *d_fam_mem = nullptr;
    bool fam_any = false;
    std::vector<int> psid(bs.shells.size(), -1);
    std::vector<void *> fam_allocs;                              // per-slab lists of the bra families (freed with the other lists of the build)
    {
        int nid = 0;
        for (size_t a = 0; a < bs.shells.size(); ++a) {
            if (psid[a] >= 0) continue;
            psid[a] = nid;
            const tf::Shell &sa = bs.shells[a];
            for (size_t b = a + 1; b < bs.shells.size(); ++b) {
                const tf::Shell &sb = bs.shells[b];
                if (psid[b] >= 0 || sb.z != sa.z || sb.L != sa.L || sb.nprim != sa.nprim || sb.ncomp != sa.ncomp || sb.full != sa.full) continue;
                bool same = true;
                for (int q = 0; q < sa.nprim && same; ++q) same = bs.s_exp[sa.prim_off + q] == bs.s_exp[sb.prim_off + q];
                for (int q = 0; q < sa.ncomp && same; ++q)
                    same = bs.c_lx[sa.comp_off + q] == bs.c_lx[sb.comp_off + q] && bs.c_ly[sa.comp_off + q] == bs.c_ly[sb.comp_off + q] &&
                           bs.c_lz[sa.comp_off + q] == bs.c_lz[sb.comp_off + q] && bs.c_scale[sa.comp_off + q] == bs.c_scale[sb.comp_off + q];
                if (same) psid[b] = nid;
            }
            ++nid;
        }
        for (int g = 0; g < NGRP; ++g) {
            fam_goff[g] = (int)fam_heads.size();
            if (!(g & 1) || fam_off) continue;                     // contracted groups only
            std::map<std::pair<int, int>, std::vector<int>> open_fam;   // key -> index of the family still taking members (in fam_ptr order)
            std::vector<std::vector<int>> fams;
            for (int k = kets_goff[g]; k < kets_goff[g + 1]; ++k) {
                const int p = kets_all_host[k];
                const auto key = std::make_pair(psid[bs.pairs[p].A], psid[bs.pairs[p].B]);
                auto it = open_fam.find(key);
                if (it == open_fam.end() || (int)fams[it->second.back()].size() >= FAM_MM) {
                    fams.emplace_back();
                    open_fam[key].push_back((int)fams.size() - 1);
                    it = open_fam.find(key);
                }
                fams[it->second.back()].push_back(p);
            }
            for (const auto &f : fams) {
                fam_heads.push_back(f[0]);
                fam_mem.insert(fam_mem.end(), f.begin(), f.end());
                fam_ptr.push_back((int)fam_mem.size());
                if (f.size() > 1) fam_any = true;
            }
        }
        fam_goff[NGRP] = (int)fam_heads.size();
        if (fam_any) {
            if ((rc = upload(ctx, fam_heads, &d_fam_heads, false)) || (rc = upload(ctx, fam_ptr, &d_fam_ptr, false)) ||
                (rc = upload(ctx, fam_mem, &d_fam_mem, false)))
                return rc;
        }
    }
    // my bra pairs ordered by class: a slab is a run of that list, launches go per (bra class run, ket class)
    std::vector<int> mine_sorted;
    for (int c = 0; c < ncls; ++c)
        for (int p : ctx->class_pairs[c])
            if (owner[p] == ctx->rank) mine_sorted.push_back(p);
    // streams so that the many small class launches of a slab overlap
    const int NSTREAM_MAX = tf_ctx::NSTREAM_MAX;
    const int NSTREAM = getenv("TF_ERI_NSTREAM") ? std::max(1, std::min(NSTREAM_MAX, atoi(getenv("TF_ERI_NSTREAM")))) : NSTREAM_MAX;
    if (!ctx->have_streams) {
        for (int k = 0; k < NSTREAM_MAX; ++k) {
            HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->streams[k], hipStreamNonBlocking));
            HIPCHK(ctx, hipEventCreateWithFlags(&ctx->sev[k], hipEventDisableTiming));
        }
        HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->cstream, hipStreamNonBlocking));
        for (auto &e : ctx->slab_done) HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto &e : ctx->slab_lists) HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->have_streams = true;
    }
    hipStream_t *streams = ctx->streams;
    hipEvent_t *sev = ctx->sev;
    int launch_count = 0;
    DBG("streams created");

    std::vector<LRec> lrecs_host;                              // per (La, Lb | Lc, Ld), filled by make_lrecs below
    hipError_t team_error = hipSuccess;                        // first failed launch of a team kernel
    static const bool team_off = getenv("TF_ERI_TEAM") && getenv("TF_ERI_TEAM")[0] == '0';
    const bool team_ok = packed && !team_off && bs.epool.size() < 0xffffffffull;
    const bool use_team_pc = team_ok && per_class;                                        // per-class mode: eri_team_kernel for the uncontracted classes
    // (off by default: on the BASELINE basis sets eri_cfact_kernel is faster -- N2/cc-pVTZ 1.2 ms against 3.3-4.2 ms, Ar2/cc-pVQZ 18 ms against
    // 22-25 ms of ERI kernels: a team walks the primitive quartets of its shell quartet one after the other, a chain of dependent phases
    // with nothing else on the chip to hide it; TF_ERI_TEAMC=1 switches it on for experiments and for the parity tests)
    static const bool teamc_off = !(getenv("TF_ERI_TEAMC") && getenv("TF_ERI_TEAMC")[0] == '1');
    const bool use_teamc = team_ok && !per_class && !teamc_off;                          // small-problem mode: eri_teamc_kernel over task lists
    const bool use_team = use_team_pc || use_teamc;                                       // the team kernels' tables are needed
    static const int teamc_pqmax = getenv("TF_TEAMC_PQMAX") ? atoi(getenv("TF_TEAMC_PQMAX")) : 700;
    struct KClassTab { int pS[5] = {0, 0, 0, 0, 0}; int nkap = 0, nnzT = 0, tp_off = 0, te_off = 0; };
    std::vector<KClassTab> kct(ncls);
    int *d_kq_ptr = nullptr, *d_kq_off = nullptr, *d_kt_ptr = nullptr, *d_kt_k = nullptr, *d_kcnt = nullptr;
    double *d_kt_c = nullptr;
    KetRec *d_ketrec = nullptr;                                // parallel to d_kets (class-sorted ket list)
    int *d_tflat = nullptr;                                    // flat component / output lists of the class pairs (TClass::flat_off)
    std::vector<int> flat_off;
    BraRec *d_brarec = nullptr;                                // parallel to the slab's bra list d_bra
    const int *d_bra_base = nullptr;
    // the class-wide part of a team kernel's class record (tf_eri_team.hip.h) for (bra pair class, ket pair class)
    auto tclass_common = [&](int bcls, int kcls, int &maxblk, int &maxK, int &nnzc) {
        const DPair &hb = ctx->host_pairs[ctx->class_pairs[bcls][0]], &hk = ctx->host_pairs[ctx->class_pairs[kcls][0]];
        const KClassTab &kt = kct[kcls];
        TClass t{};
        t.La = hb.La; t.Lb = hb.Lb; t.Lc = hk.La; t.Ld = hk.Lb;
        t.nTab = (t.La + 1) * (t.Lb + 1); t.nTcd = (t.Lc + 1) * (t.Ld + 1); t.nT = t.nTab * t.nTcd;
        t.inv_nTcd = 1.0f / (float)t.nTcd;
        t.nab = hb.nca * hb.ncb; t.ncd = hk.nca * hk.ncb;
        maxblk = 1; maxK = 1; nnzc = 0;
        for (int i = 0; i < 5; ++i) { t.pA[i] = hb.pcls[i]; t.pK[i] = hk.pcls[i]; t.pS[i] = kt.pS[i]; }
        for (int i = 0; i < 4; ++i) {
            maxK = std::max(maxK, t.pK[i + 1] - t.pK[i]);
            maxblk = std::max(maxblk, (t.pA[i + 1] - t.pA[i]) * (t.pK[i + 1] - t.pK[i]));
            nnzc += (t.pA[i + 1] - t.pA[i]) * (t.pK[i + 1] - t.pK[i]);
        }
        t.nkap = kt.nkap; t.nnzT = kt.nnzT; t.tabA = hb.tab_off; t.tabK = hk.tab_off; t.ktp_off = kt.tp_off; t.kte_off = kt.te_off;
        t.nEab = hb.nE; t.nEcd = hk.nE; t.RLS = H.RLS; t.nacc = nnzc;
        t.nout = 0;
        for (int i = 0; i < 4; ++i) {
            t.invK[i] = 1.0f / (float)std::max(1, t.pK[i + 1] - t.pK[i]);
            t.invS[i] = 1.0f / (float)std::max(1, t.pS[i + 1] - t.pS[i]);
            t.nout += (t.pA[i + 1] - t.pA[i]) * (t.pS[i + 1] - t.pS[i]);
        }
        return t;
    };
    // bra_Amax: largest first shell among the bra pairs of the run -- in the packed layout only kets with first shell <= it are needed
    // (the class ket lists ascend in the first shell, so that is a prefix: workgroups beyond it are not even launched)
    auto class_launch = [&](int bcls, int kcls, int max_npp_bra, unsigned n_bra, const int *d_bra, const long long *d_braoff, int bra_Amax) {
        const tf::Pair &pb = bs.pairs[ctx->class_pairs[bcls][0]], &pk = bs.pairs[ctx->class_pairs[kcls][0]];
        const tf::Shell &sa = bs.shells[pb.A], &sb = bs.shells[pb.B], &sc = bs.shells[pk.A], &sd = bs.shells[pk.B];
        QClass q{};
        q.La = pb.La; q.Lb = pb.Lb; q.Lc = pk.La; q.Ld = pk.Lb;
        q.L = q.La + q.Lb + q.Lc + q.Ld;
        q.tsize = (q.L + 1) * (q.L + 2) / 2;
        q.nca = sa.ncomp; q.ncb = sb.ncomp; q.ncc = sc.ncomp; q.ncd = sd.ncomp;
        q.ncomp = q.nca * q.ncb * q.ncc * q.ncd;
        q.npp_ab = max_npp_bra; q.npp_cd = cls_maxnpp[kcls]; q.npq = q.npp_ab * q.npp_cd;      // class maxima (LDS sizing)
        q.nEab = pb.nE; q.nEcd = pk.nE;
        q.n_ket = ket_off[kcls + 1] - ket_off[kcls];
        if (packed) {
            const std::vector<int> &kl = ctx->class_pairs[kcls];
            q.n_ket = (int)(std::upper_bound(kl.begin(), kl.end(), bra_Amax, [&](int a, int p) { return a < bs.pairs[p].A; }) - kl.begin());
            if (q.n_ket == 0) return;
        }
        q.fused = 1; q.spherical = spherical ? 1 : 0;
        q.nsc = spherical ? sc.nsph : sc.ncomp; q.nsd = spherical ? sd.nsph : sd.ncomp;
        q.Nout = N; q.ld = ld;
        q.tri = packed ? 1 : 0;
        double *d_out_slab = d_T2;                               // fused kernels write the half-transformed slab directly
        const int *d_ket = d_kets + ket_off[kcls];
        hipStream_t st = streams[launch_count++ % NSTREAM];
        // TF_ERI_CLASS_TIMES=1 (diagnostic): every class launch alone on the device, its time printed with the class
        static const bool class_times = getenv("TF_ERI_CLASS_TIMES") != nullptr;
        struct ClassTimer {
            bool on; const QClass &q; unsigned nb; std::chrono::steady_clock::time_point t0;
            ClassTimer(bool o, const QClass &qq, unsigned n) : on(o), q(qq), nb(n) { if (on) { (void)hipDeviceSynchronize(); t0 = std::chrono::steady_clock::now(); } }
            ~ClassTimer() {
                if (!on) return;
                (void)hipDeviceSynchronize();
                const double ms = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                fprintf(stderr, "[tf eri class] (%d %d|%d %d) npq %d bra %u ket %d quartets %.0f: %.3f ms, %.1f ns per quartet\n", q.La, q.Lb, q.Lc, q.Ld,
                        q.npq, nb, q.n_ket, (double)nb * q.n_ket, ms, 1e6 * ms / ((double)nb * q.n_ket));
            }
        } class_timer(class_times, q, n_bra);
        if (use_team_pc && q.npq == 1 && q.La + q.Lb <= TF_TEAM_LMAX && q.Lc + q.Ld <= TF_TEAM_LMAX) {
            // one shell quartet per team of lanes, tables private to the team (tf_eri_team.hip.h)
            int maxblk = 1, maxK = 1, nnzc = 0;
            TClass t = tclass_common(bcls, kcls, maxblk, maxK, nnzc);
            t.n_ket = q.n_ket;
            const int LAB = q.La + q.Lb, LCD = q.Lc + q.Ld, NM = q.L / 2 + 1, XS = NM | 1, RSr = q.L + 2;
            auto even = [](int x) { return (x + 1) & ~1; };
            int o = 0;
            t.oE12 = o; o += even(2 * t.nEab);
            t.oOffA = o; o += 2 * t.nab;
            t.oScA = o; o += even(t.nab);
            t.oOffK = o; o += 2 * t.ncd;
            t.oTp = o; o += even((t.nkap + 2) / 2);
            t.oTk = o; o += even((t.nnzT + 1) / 2);
            t.oTc = o; o += even(t.nnzT);
            t.oRowOff = o; o += t.nab;
            const int foff = flat_off.empty() ? -1 : flat_off[(size_t)bcls * ncls + kcls];
            auto vmax_of = [](int tm) { return tm == 256 ? 2048 : (tm == 64 ? 512 : 96); };
            const int nacc_pad = even(nnzc);
            t.oCompW = o; t.oOutW = o + nacc_pad / 2; t.flat_off = std::max(0, foff); t.nflat = nacc_pad + 2 * t.nout;
            const int flat_doubles = even((t.nflat + 1) / 2);
            const int shared_noflat = o;
            t.shared_doubles = o;
            const int nG = t.nTcd * (LAB + 1) * NM, scr1 = 2 * t.nEcd + (q.L + 1) * RSr + nG;
            // (a class whose parity-allowed components fit the team's block runs the flat lists: one loop over all components)
            auto flat_for = [&](int tm) { return foff >= 0 && nnzc <= vmax_of(tm); };
            auto vcap_of = [&](int tm) { return flat_for(tm) ? even(std::max(scr1, nnzc)) : even(std::max(scr1, std::max(maxK, std::min(maxblk, vmax_of(tm))))); };
            auto team_doubles_of = [&](int tm) { return 2 * t.nT * XS + vcap_of(tm) + even((t.nkap + 1) / 2); };
            auto bytes_of = [&](int tm) { return ((size_t)shared_noflat + (flat_for(tm) ? flat_doubles : 0) + (size_t)(256 / tm) * team_doubles_of(tm)) * sizeof(double); };
            // lanes per quartet: 16 for the smallest classes; a wave while four quartets' tables fit about half of the LDS (two workgroups per
            // CU); the whole workgroup beyond.  (Measured at N = 400, ERI kernels: 36 / 52 / 76 / 100 KB limit: 35.3 / 32.8 / 31.9 / 33 ms.)
            static const int force_team = getenv("TF_ERI_TEAM_SIZE") ? atoi(getenv("TF_ERI_TEAM_SIZE")) : 0;
            static const int lds_kb = getenv("TF_TEAM_LDS_KB") ? atoi(getenv("TF_TEAM_LDS_KB")) : 76;
            static const int t16_nnz = getenv("TF_TEAM16_NNZ") ? atoi(getenv("TF_TEAM16_NNZ")) : 96;
            static const long long kpw_div = getenv("TF_TEAM_KPW_DIV") ? atoll(getenv("TF_TEAM_KPW_DIV")) : 2048;
            static const long long kpw_max = getenv("TF_TEAM_KPW_MAX") ? atoll(getenv("TF_TEAM_KPW_MAX")) : 16;
            int team = 0;
            if (t.nT <= 16 && nnzc <= t16_nnz && eri_team_available(LAB, LCD, 16)) team = 16;
            else if (eri_team_available(LAB, LCD, 64) && bytes_of(64) <= (size_t)lds_kb * 1024) team = 64;
            else if (eri_team_available(LAB, LCD, 256) && bytes_of(256) <= 160 * 1024 - 256) team = 256;
            else if (eri_team_available(LAB, LCD, 64) && bytes_of(64) <= 160 * 1024 - 256) team = 64;
            if (force_team && eri_team_available(LAB, LCD, force_team) && bytes_of(force_team) <= 160 * 1024 - 256) team = force_team;
            if (team) {
                const int NT = 256 / team;
                t.vcap = vcap_of(team); t.team_doubles = team_doubles_of(team);
                t.flat = flat_for(team) ? 1 : 0;
                t.shared_doubles = shared_noflat + (t.flat ? flat_doubles : 0);
                // a workgroup walks over several ket groups (shared staging once): enough workgroups to fill the chip, at most 16 groups each
                // (measured at N = 400, ERI kernels: >= 65536 / 16384 / 4096 / 2048 / 1024 / 512 workgroups per launch aimed at:
                // 36.0 / 32.8 / 28.3 / 26.8 / 26.5-27.1 / 27.9 ms)
                const long long groups = (q.n_ket + NT - 1) / NT;
                const long long kpw = std::max<long long>(1, std::min<long long>(kpw_max, groups * n_bra / kpw_div));
                TeamLaunch a{LAB, LCD, team, dim3((unsigned)((groups + kpw - 1) / kpw), n_bra), bytes_of(team), st, &ctx->db, &t,
                             d_brarec + (d_bra - d_bra_base), d_ketrec + ket_off[kcls], d_kcnt + (size_t)kcls * nsh, d_out_slab};
                const hipError_t e = eri_team_launch(a);
                if (e != hipSuccess) { team_error = e; }
                return;
            }
        }
        if (q.npq == 1 && q.ncomp <= 128) {
            // several uncontracted shell quartets per workgroup
            int ncp = 1;
            while (ncp < q.ncomp) ncp <<= 1;
            q.ncp = ncp;
            q.G = std::max(1, std::min(TF_ERI_THREADS / ncp, TF_ERI_THREADS / (q.L + 1)));
            q.PB = q.G; q.stride = q.G | 1;
            const int kc = q.ncc + q.ncd;
            int o = 0;
            q.offR = o; o += q.stride * q.tsize;
            q.offPref = o; o += q.G;
            q.offPQ = o; o += q.G;
            q.offRed = o;
            q.offEab = o; o += 2 * q.nEab;
            q.offEcd = o; o += q.G * 2 * q.nEcd;
            q.offScale = o; o += 42 + kc * q.G;
            q.offLmn = o; o += (42 + kc * q.G + q.G + 1) / 2;
            q.offBlk = o; o += q.G * q.ncomp;
            q.offCsr = o; o += TF_CSR_DOUBLES;
            q.lds_doubles = o;
            const dim3 grid((q.n_ket + q.G - 1) / q.G, n_bra);
            hipLaunchKernelGGL(eri_multi_kernel, grid, dim3(TF_ERI_THREADS), (size_t)o * sizeof(double), st, ctx->db, q, d_bra, d_braoff,
                               d_ket, Nc, d_out_slab);
        } else if (q.npq == 1 && (q.La + 1) * (q.Lb + 1) * (q.Lc + 1) * (q.Ld + 1) * 2 * (q.L / 2 + 1) <= 7000 && !getenv("TF_ERI_NOFACT")) {
            // uncontracted, many components: per-axis factor tables in LDS
            const int nT = (q.La + 1) * (q.Lb + 1) * (q.Lc + 1) * (q.Ld + 1), nM = q.L / 2 + 1;
            q.PB = 1; q.stride = 1; q.G = 1; q.ncp = 0;
            const LRec &lr = lrecs_host[((q.La * 6 + q.Lb) * 6 + q.Lc) * 6 + q.Ld];
            q.tupG_off = lr.tupG_off; q.tupXZ_off = lr.tupXZ_off;
            int o = 0;
            q.offR = o; o += q.tsize;
            q.offPref = o; o += 2;
            q.offPQ = o; o += 2;
            q.offEab = o; o += 2 * q.nEab;
            q.offEcd = o; o += 2 * q.nEcd;
            const int tables_end = o;                            // R, prefactors and E tables: dead once X and Z are built
            q.offScale = o; o += 84;
            q.offLmn = o; o += 42;
            q.offRed = o; o += 2 * nT * nM;                      // X and Z tables
            const int nG = (q.Lc + 1) * (q.Ld + 1) * (q.La + q.Lb + 1) * nM;
            q.offBlk = o; o += std::max((int)TF_BLK_DOUBLES, nG);   // the ket half of the z tables lives here until the components start
            q.offG = q.offBlk;
            q.offTab = o; o += 2 * (q.nca * q.ncb + q.ncc * q.ncd) + 2;
            if (tables_end >= TF_CSR_DOUBLES) q.offCsr = 0;      // staged over the dead tables (after the X/Z barrier)
            else { q.offCsr = o; o += TF_CSR_DOUBLES; }
            q.lds_doubles = o;
            // (TF_ERI_FACT_THREADS=64|128: smaller workgroups for experiments -- measured slower at N = 400, 127 / 159 ms against 113-127 ms:
            // the LDS of a quartet limits the workgroups per CU, so fewer waves per workgroup are fewer waves per CU)
            static const int force_thr = getenv("TF_ERI_FACT_THREADS") ? atoi(getenv("TF_ERI_FACT_THREADS")) : 0;
            const int fact_threads = (force_thr == 64 || force_thr == 128) ? force_thr : TF_ERI_THREADS;
            hipLaunchKernelGGL(eri_fact_kernel, dim3(q.n_ket, n_bra), dim3(fact_threads), (size_t)o * sizeof(double), st, ctx->db, q, d_bra,
                               d_braoff, d_ket, Nc, d_out_slab);
        } else {
            const int RB = 3584, EB = 3072;                     // LDS doubles for R tables / staged E tables
            int PB = RB / q.tsize - 1;
            PB = std::max(1, std::min(std::min(PB, TF_ERI_THREADS), q.npq));
            q.PB = PB; q.stride = PB | 1; q.G = 1; q.ncp = 0;
            const int needE = q.npp_ab * 2 * q.nEab + q.npp_cd * 2 * q.nEcd;
            const bool stage = needE <= EB;
            int o = 0;
            q.offR = o; o += q.stride * q.tsize;
            q.offPref = o; o += TF_ERI_THREADS;
            q.offPQ = o; o += TF_ERI_THREADS;
            q.offRed = o; o += TF_ERI_THREADS;
            q.offEab = o; o += stage ? q.npp_ab * 2 * q.nEab : 0;
            q.offEcd = o; o += stage ? q.npp_cd * 2 * q.nEcd : 0;
            q.offScale = o; o += 84;
            q.offLmn = o; o += 42;
            q.offBlk = o; o += TF_BLK_DOUBLES;
            q.offCsr = o; o += TF_CSR_DOUBLES;
            q.lds_doubles = o;
            const dim3 grid(q.n_ket, n_bra);
            if (stage)
                hipLaunchKernelGGL((eri_class_kernel<true, false>), grid, dim3(TF_ERI_THREADS), (size_t)o * sizeof(double), st, ctx->db, q,
                                   d_bra, d_braoff, d_ket, Nc, d_out_slab);
            else
                hipLaunchKernelGGL((eri_class_kernel<false, false>), grid, dim3(TF_ERI_THREADS), (size_t)o * sizeof(double), st, ctx->db, q,
                                   d_bra, d_braoff, d_ket, Nc, d_out_slab);
        }
    };

    DBG("stage: small-problem groups");
    // Small problems: per slab one launch per (bra group, ket group), each mixing the classes of its groups (LDS carved by the
    // capacities the groups need).  bra_host: the slab's bra pairs (sorted by group).
    static const bool old_generic = getenv("TF_ERI_GENERIC_OLD") != nullptr;
    auto generic_launch_old = [&](unsigned n_bra, const int *d_bra, const long long *d_braoff, unsigned n_ket, const int *d_ket, hipStream_t st) {
        QClass q{};
        const int RB = 2048, EBa = 1024, EBc = 1024;       // (doubling the E capacities halves the occupancy: Ar2 build 0.084 -> 0.18 s)
        int o = 0;
        q.offR = o; o += RB;
        q.offPref = o; o += TF_ERI_THREADS;
        q.offPQ = o; o += TF_ERI_THREADS;
        q.offRed = o; o += TF_ERI_THREADS;
        q.offEab = o; o += EBa;
        q.offEcd = o; o += EBc;
        q.offScale = o; o += 84;
        q.offLmn = o; o += 42;
        q.lds_doubles = o;
        q.offBlk = o;                                            // unused (unfused)
        q.G = 1; q.n_ket = (int)n_ket; q.fused = 0;
        q.tri = packed ? 1 : 0;
        hipLaunchKernelGGL((eri_class_kernel<true, true>), dim3(n_ket, n_bra), dim3(TF_ERI_THREADS), (size_t)o * sizeof(double), st,
                           ctx->db, q, d_bra, d_braoff, d_ket, Nc, d_C);
    };
    struct GroupStat { int maxLp = 0, maxT = 1, maxLp1 = 1, maxE = 0, maxcomp = 1, maxnpp = 1; };
    auto group_stat = [&](const int *pairs_host, size_t n) {
        GroupStat g;
        for (size_t k = 0; k < n; ++k) {
            const tf::Pair &pr = bs.pairs[pairs_host[k]];
            g.maxLp = std::max(g.maxLp, pr.La + pr.Lb);
            g.maxT = std::max(g.maxT, (pr.La + 1) * (pr.Lb + 1));
            g.maxE = std::max(g.maxE, pr.npp * 2 * pr.nE);
            g.maxcomp = std::max(g.maxcomp, bs.shells[pr.A].ncomp * bs.shells[pr.B].ncomp);
            g.maxnpp = std::max(g.maxnpp, pr.npp);
        }
        g.maxLp1 = g.maxLp + 1;
        return g;
    };
    DBG("stage: carve-outs");
    // LDS carve-out of the launch (bra group gb, ket group gk), from the largest angular momenta / contraction depths of the groups
    CFCaps gcaps[NGRP][NGRP];
    bool gcaps_fit[NGRP][NGRP], gcaps_gtab[NGRP][NGRP];
    auto make_caps = [&]() {
        GroupStat gs[NGRP];
        for (int g = 0; g < NGRP; ++g) gs[g] = group_stat(kets_all_host.data() + kets_goff[g], (size_t)(kets_goff[g + 1] - kets_goff[g]));
        for (int gb = 0; gb < NGRP; ++gb)
            for (int gk = 0; gk < NGRP; ++gk) {
                const GroupStat &sb = gs[gb], &sk = gs[gk];
                const int Lmax = sb.maxLp + sk.maxLp, nM = Lmax / 2 + 1, tsize = (Lmax + 1) * (Lmax + 2) / 2;
                const int xz = sb.maxT * sk.maxT * nM, gsz = sk.maxT * sb.maxLp1 * nM;
                const bool deep = sb.maxnpp > 1 || sk.maxnpp > 1;
                // batch size aimed at and LDS doubles for the X / Z tables and the staged Hermite tables (tuning knobs; smaller carve-outs
                // mean more workgroups per CU: Ar2/cc-pVQZ ERI kernels 21.1 ms with 48 / 1536 / 1536, 18.3 ms with 24 / 768 / 1024)
                static const int k_nbt = getenv("TF_CF_NBT") ? atoi(getenv("TF_CF_NBT")) : 24, k_xz = getenv("TF_CF_XZ") ? atoi(getenv("TF_CF_XZ")) : 768;
                static const int k_e = getenv("TF_CF_E") ? atoi(getenv("TF_CF_E")) : 1024;
                const int nbt = deep ? std::min(TF_ERI_THREADS / (Lmax + 1), k_nbt) : 1;       // primitive quartets per batch aimed at
                CFCaps c{};
                int o = 0;
                c.offR = o; c.capR = std::max(2 * tsize, std::min((nbt + 1) * tsize, 2 * k_xz * 2 / 3)); o += c.capR;
                c.offPref = o; o += TF_ERI_THREADS;
                c.offPQ = o; o += TF_ERI_THREADS;
                c.offPP = o; o += TF_ERI_THREADS;
                c.offG = o; c.capG = std::max(gsz, std::min(nbt * gsz, k_xz * 2 / 3)); o += c.capG;
                c.capXZ = std::max(xz, std::min(nbt * xz, k_xz));
                c.offX = o; o += c.capXZ;
                c.offZ = o; o += c.capXZ;
                c.offTupG = o; o += (gsz + 3) / 4;
                c.offTupXZ = o; o += (xz + 3) / 4;
                c.offEab = o; c.capEab = std::min(sb.maxE, k_e); o += c.capEab;
                c.offEcd = o; c.capEcd = std::min(sk.maxE, k_e); o += c.capEcd;
                c.offRed = o; o += TF_ERI_THREADS;
                c.offKm = o; c.capKm = fam_off ? 0 : FAM_MM * (((gk & 1) ? sk.maxnpp : 0) + ((gb & 1) ? sb.maxnpp : 0)); o += c.capKm;
                c.lds_doubles = o;
                c.tri = packed ? 1 : 0;
                c.dbg_npq_lo = 0; c.dbg_npq_hi = 0x7fffffff;
                c.team_lmax = use_teamc ? TF_TEAM_LMAX : -1; c.team_pqmax = teamc_pqmax;
                if (const char *e = getenv("TF_ERI_DBG_NPQ")) (void)sscanf(e, "%d:%d", &c.dbg_npq_lo, &c.dbg_npq_hi);
                gcaps[gb][gk] = c;
                gcaps_fit[gb][gk] = (size_t)o * sizeof(double) <= 160 * 1024 - 256;
                gcaps_gtab[gb][gk] = false;
                if (!gcaps_fit[gb][gk] && !deep) {
                    // (hh|hh)-sized tables: G, X and Z of a workgroup go to global memory, everything else stays in LDS
                    CFCaps d = c;
                    int q = 0;
                    d.offR = q; q += d.capR;
                    d.offPref = q; q += TF_ERI_THREADS;
                    d.offPQ = q; q += TF_ERI_THREADS;
                    d.offPP = q; q += TF_ERI_THREADS;
                    d.offTupG = q; q += (gsz + 3) / 4;
                    d.offTupXZ = q; q += (xz + 3) / 4;
                    d.offEab = q; q += d.capEab;
                    d.offEcd = q; q += d.capEcd;
                    d.offRed = q; q += TF_ERI_THREADS;
                    d.offKm = q; q += d.capKm;
                    d.lds_doubles = q;
                    d.capG = gsz; d.capXZ = xz;
                    d.offG = 0; d.offX = gsz; d.offZ = gsz + xz;
                    d.gtab_doubles = gsz + 2 * xz;
                    if ((size_t)q * sizeof(double) <= 160 * 1024 - 256) { gcaps[gb][gk] = d; gcaps_fit[gb][gk] = true; gcaps_gtab[gb][gk] = true; }
                }
            }
    };
    // per (La, Lb | Lc, Ld): table sizes, index words of the table entries, batch capacity under the caps of its launch
    auto make_lrecs = [&](bool with_caps) -> int {
        std::vector<LRec> recs(6 * 6 * 6 * 6, LRec{});
        std::vector<unsigned short> tup;
        std::vector<char> seen_pair(36, 0);
        for (const tf::Pair &pr : bs.pairs) seen_pair[pr.La * 6 + pr.Lb] = 1;
        auto grp_of = [&](int lp) { return lp_group(lp); };
        for (int ab = 0; ab < 36; ++ab)
            for (int cd = 0; cd < 36; ++cd) {
                if (!seen_pair[ab] || !seen_pair[cd]) continue;
                const int La = ab / 6, Lb = ab % 6, Lc = cd / 6, Ld = cd % 6;
                const int gb = grp_of(La + Lb), gk = grp_of(Lc + Ld);
                LRec r{};
                r.L = La + Lb + Lc + Ld; r.nM = r.L / 2 + 1; r.tsize = (r.L + 1) * (r.L + 2) / 2;
                r.nT = (La + 1) * (Lb + 1) * (Lc + 1) * (Ld + 1); r.xz = r.nT * r.nM;
                const int Lab1 = La + Lb + 1;
                r.gsz = (Lc + 1) * (Ld + 1) * Lab1 * r.nM;
                r.lgG = 0; while ((1 << r.lgG) < r.gsz && (1 << r.lgG) < TF_ERI_THREADS) ++r.lgG;
                r.lgX = 0; while ((1 << r.lgX) < r.xz && (1 << r.lgX) < TF_ERI_THREADS) ++r.lgX;
                // batch capacity: the smallest over the launches (contracted bra and / or ket group) this tuple can occur in
                int nb = TF_ERI_THREADS / (r.L + 1);
                for (int fb = 0; fb < 2 && with_caps; ++fb)
                    for (int fk = 0; fk < 2; ++fk) {
                        if (fb + fk == 0) continue;
                        const CFCaps &c = gcaps[2 * gb + fb][2 * gk + fk];
                        if (c.capR < 2 * r.tsize || c.capG < r.gsz || c.capXZ < r.xz) continue;   // (no such quartet in that launch)
                        nb = std::min(nb, c.capR / r.tsize - 1);
                        nb = std::min(nb, std::min(c.capG / r.gsz, c.capXZ / r.xz));
                    }
                r.nb_cap = std::max(nb, 1);
                // entry e of the G table = ((cc (Ld + 1) + d) Lab1 + v) nM + n, of the X / Z tables = (((a2 (Lb + 1) + b2) (Lc + 1) + cc) (Ld + 1) + d) nM + m
                // (nested loops in that order: a cc-pVQZ basis has 625 tuples and 6e5 entries -- with a division chain per entry this was
                // 2 ms of every tensor build)
                r.tupG_off = (int)tup.size();
                tup.resize(tup.size() + (size_t)r.gsz + (size_t)r.xz);
                unsigned short *tp = tup.data() + r.tupG_off;
                for (int cc = 0; cc <= Lc; ++cc)
                    for (int d = 0; d <= Ld; ++d)
                        for (int v = 0; v < Lab1; ++v)
                            for (int n = 0; n < r.nM; ++n) *tp++ = (unsigned short)(n | (v << 4) | (d << 8) | (cc << 11));
                r.tupXZ_off = r.tupG_off + r.gsz;
                for (int a2 = 0; a2 <= La; ++a2)
                    for (int b2 = 0; b2 <= Lb; ++b2)
                        for (int cc = 0; cc <= Lc; ++cc)
                            for (int d = 0; d <= Ld; ++d)
                                for (int m = 0; m < r.nM; ++m) *tp++ = (unsigned short)(m | (a2 << 4) | (b2 << 7) | (cc << 10) | (d << 13));
                recs[ab * 36 + cd] = r;
            }
        if (ctx->d_lrec) { (void)tf_free(ctx->d_lrec); ctx->d_lrec = nullptr; }
        if (ctx->d_tup) { (void)tf_free(ctx->d_tup); ctx->d_tup = nullptr; }
        int rc2;
        if ((rc2 = upload(ctx, recs, &ctx->d_lrec, false)) || (rc2 = upload(ctx, tup, &ctx->d_tup, false))) return rc2;
        ctx->db.lrec = ctx->d_lrec; ctx->db.tup = ctx->d_tup;
        lrecs_host = recs;
        return TF_OK;
    };
    DBG("stage: launch lambda defined");
    // One launch per (bra group, ket group) with work.  A process has few hardware queues (4 by default) and the launches of one
    // queue run one after the other, each as long as its slowest workgroup: the launches are spread over NQ streams by estimated
    // cost (heaviest first, always onto the least loaded stream) instead of round-robin over all of them.
    auto generic_launch = [&](const std::vector<int> &bra_host, const int *d_bra, const long long *d_braoff) -> int {
        struct Launch { size_t b0, b1; int gb, gk; double cost; };
        std::vector<Launch> launches;
        double ket_cost[NGRP];
        for (int g = 0; g < NGRP; ++g) {
            ket_cost[g] = 0.0;
            for (int k = kets_goff[g]; k < kets_goff[g + 1]; ++k) ket_cost[g] += (double)pair_cost(kets_all_host[k]);
        }
        for (size_t b0 = 0; b0 < bra_host.size();) {
            const int gb = pair_group(bra_host[b0]);
            size_t b1 = b0;
            double bc = 0.0;
            while (b1 < bra_host.size() && pair_group(bra_host[b1]) == gb) { bc += (double)pair_cost(bra_host[b1]); ++b1; }
            for (int gk = 0; gk < NGRP; ++gk)
                if (kets_goff[gk + 1] > kets_goff[gk]) launches.push_back(Launch{b0, b1, gb, gk, bc * ket_cost[gk]});
            b0 = b1;
        }
        std::stable_sort(launches.begin(), launches.end(), [](const Launch &x, const Launch &y) { return x.cost > y.cost; });
        // (as many streams as the process has hardware queues: 4 unless GPU_MAX_HW_QUEUES says otherwise -- tuna_amd sets 16)
        const int hwq = getenv("GPU_MAX_HW_QUEUES") ? std::max(1, atoi(getenv("GPU_MAX_HW_QUEUES"))) : 4;
        const int NQ = std::min(NSTREAM, getenv("TF_ERI_NQ") ? std::max(1, atoi(getenv("TF_ERI_NQ"))) : std::max(4, std::min(8, hwq)));
        std::vector<double> load(NQ, 0.0);
        struct BraFam { int *d_ptr = nullptr, *d_mem = nullptr; unsigned n = 0; };
        std::map<std::pair<size_t, int>, BraFam> bra_fams;       // by (first position of the run of bra pairs, largest family)
        // families of the bra pairs [b0, b1) of the slab's list, at most `most` members each: built and uploaded once per (slab, bra group)
        auto bra_families = [&](size_t b0, size_t b1, int most) -> const BraFam * {
            auto bf = bra_fams.find(std::make_pair(b0, most));
            if (bf != bra_fams.end()) return &bf->second;
            BraFam nf;
            std::vector<int> bptr{0}, bmem;
            std::map<std::pair<int, int>, int> open_fam;
            std::vector<std::vector<int>> fams;
            for (size_t y = b0; y < b1; ++y) {
                const tf::Pair &pr = bs.pairs[bra_host[y]];
                const auto key = std::make_pair(psid[pr.A], psid[pr.B]);
                auto it = open_fam.find(key);
                if (it == open_fam.end() || (int)fams[it->second].size() >= most) {
                    fams.emplace_back();
                    open_fam[key] = (int)fams.size() - 1;
                    it = open_fam.find(key);
                }
                fams[it->second].push_back((int)y);
            }
            for (const auto &f : fams) { bmem.insert(bmem.end(), f.begin(), f.end()); bptr.push_back((int)bmem.size()); }
            if (upload(ctx, bptr, &nf.d_ptr, false) || upload(ctx, bmem, &nf.d_mem, false)) return nullptr;
            fam_allocs.push_back(nf.d_ptr); fam_allocs.push_back(nf.d_mem);
            nf.n = (unsigned)(bptr.size() - 1);
            return &bra_fams.emplace(std::make_pair(b0, most), nf).first->second;
        };
        for (const Launch &l : launches) {
            const int qi = (int)(std::min_element(load.begin(), load.end()) - load.begin());
            load[qi] += l.cost;
            hipStream_t st = streams[qi];
            const int gb = l.gb, gk = l.gk, nk = kets_goff[gk + 1] - kets_goff[gk];
            const size_t b0 = l.b0, b1 = l.b1;
            const CFCaps &c = gcaps[gb][gk];
            const size_t bytes = (size_t)c.lds_doubles * sizeof(double);
            if (old_generic || !gcaps_fit[gb][gk]) {             // the component-per-lane kernel
                generic_launch_old((unsigned)(b1 - b0), d_bra + b0, d_braoff + b0, (unsigned)nk, d_kets_all + kets_goff[gk], st);
                continue;
            }
            if (bytes > 64 * 1024 && !ctx->cfact_lds_set) {
                HIPCHK(ctx, hipFuncSetAttribute((const void *)eri_cfact_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
                HIPCHK(ctx, hipFuncSetAttribute((const void *)eri_cfact_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
                HIPCHK(ctx, hipFuncSetAttribute((const void *)eri_cfact_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
                ctx->cfact_lds_set = 160 * 1024;
            }
            if (gcaps_gtab[gb][gk]) {
                const size_t need = (size_t)nk * (b1 - b0) * (size_t)c.gtab_doubles * sizeof(double);
                if (need > ctx->gtab_bytes) {
                    HIPCHK(ctx, hipDeviceSynchronize());           // (earlier launches of this build may still use the old block)
                    if (ctx->d_gtab) (void)tf_free(ctx->d_gtab);
                    ctx->d_gtab = nullptr; ctx->gtab_bytes = 0;
                    HIPCHK(ctx, tf_malloc((void **)&ctx->d_gtab, need));
                    ctx->gtab_bytes = need;
                }
                // launches that share the block must not overlap: they all go to one stream
                hipLaunchKernelGGL((eri_cfact_kernel<true, true>), dim3((unsigned)nk, (unsigned)(b1 - b0)), dim3(TF_ERI_THREADS), bytes, streams[0], ctx->db, c,
                                   d_bra + b0, d_braoff + b0, d_kets_all + kets_goff[gk], Nc, d_C, ctx->d_gtab);
                load[qi] -= l.cost; load[0] += l.cost;
            } else if (((gb | gk) & 1) == 0)                     // both groups uncontracted: one primitive quartet per shell quartet
                hipLaunchKernelGGL(eri_cfact_kernel<true>, dim3((unsigned)nk, (unsigned)(b1 - b0)), dim3(TF_ERI_THREADS), bytes, st, ctx->db, c,
                                   d_bra + b0, d_braoff + b0, d_kets_all + kets_goff[gk], Nc, d_C);
            else if ((gb & 1) && !(gk & 1) && bra_fam_on) {       // contracted bras against uncontracted kets: (family of bra pairs, ket pair).
                // Off by default (TF_ERI_BRA_FAMILIES=1): the packed layout evaluates the kets whose first shell does not exceed the bra's,
                // and the contracted shells come first on each atom -- the contracted pairs sit on the ket side; measured on Ar2/cc-pVQZ these
                // bra families cost 5 % (fewer, longer workgroups) where the ket families gain 24 %.
                const BraFam *bf = bra_families(b0, b1, FAM_MM);
                if (!bf) return TF_ENOMEM;
                static bool bfam_attr_set = false;
                if (bytes > 64 * 1024 && !bfam_attr_set) {
                    HIPCHK(ctx, hipFuncSetAttribute((const void *)eri_cfact_kernel<false, false, FAM_MM, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
                    bfam_attr_set = true;
                }
                hipLaunchKernelGGL((eri_cfact_kernel<false, false, FAM_MM, 1>), dim3((unsigned)nk, bf->n), dim3(TF_ERI_THREADS), bytes, st,
                                   ctx->db, c, d_bra, d_braoff, d_kets_all + kets_goff[gk], Nc, d_C, (double *)nullptr, (const int *)nullptr,
                                   (const int *)nullptr, bf->d_ptr, bf->d_mem);
            }
            else if ((gb & 1) && (gk & 1) && fam_any && !cc_fam_off) {
                // contracted against contracted: families on both sides -- up to 3 bra pairs x up to 9 ket pairs per workgroup (27 accumulators
                // per component: the deepest quartets, (s13 s13|s13 s13) and friends, spend their time in the tables of 28 561 primitive quartets)
                const BraFam *bf = bra_families(b0, b1, FAM_MA_CC);
                if (!bf) return TF_ENOMEM;
                static bool ccfam_attr_set = false;
                if (bytes > 64 * 1024 && !ccfam_attr_set) {
                    HIPCHK(ctx, hipFuncSetAttribute((const void *)eri_cfact_kernel<false, false, FAM_MA_CC, FAM_MM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
                    ccfam_attr_set = true;
                }
                hipLaunchKernelGGL((eri_cfact_kernel<false, false, FAM_MA_CC, FAM_MM>), dim3((unsigned)(fam_goff[gk + 1] - fam_goff[gk]), bf->n),
                                   dim3(TF_ERI_THREADS), bytes, st, ctx->db, c, d_bra, d_braoff, d_fam_heads + fam_goff[gk], Nc, d_C, (double *)nullptr,
                                   d_fam_ptr + fam_goff[gk], d_fam_mem, bf->d_ptr, bf->d_mem);
            }
            else if ((gk & 1) && fam_any) {                       // contracted kets: one workgroup per (bra pair, family of ket pairs)
                static bool fam_attr_set = false;                 // (per process and device: the attribute belongs to the function)
                if (bytes > 64 * 1024 && !fam_attr_set) {
                    HIPCHK(ctx, hipFuncSetAttribute((const void *)eri_cfact_kernel<false, false, 1, FAM_MM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
                    fam_attr_set = true;
                }
                hipLaunchKernelGGL((eri_cfact_kernel<false, false, 1, FAM_MM>), dim3((unsigned)(fam_goff[gk + 1] - fam_goff[gk]), (unsigned)(b1 - b0)),
                                   dim3(TF_ERI_THREADS), bytes, st, ctx->db, c, d_bra + b0, d_braoff + b0, d_fam_heads + fam_goff[gk], Nc, d_C,
                                   (double *)nullptr, d_fam_ptr + fam_goff[gk], d_fam_mem);
            } else
                hipLaunchKernelGGL(eri_cfact_kernel<false>, dim3((unsigned)nk, (unsigned)(b1 - b0)), dim3(TF_ERI_THREADS), bytes, st, ctx->db, c,
                                   d_bra + b0, d_braoff + b0, d_kets_all + kets_goff[gk], Nc, d_C);
            ++launch_count;
        }
        return TF_OK;
    };
    if (!per_class) make_caps();
    DBG("stage: index-word tables");
    if ((rc = make_lrecs(!per_class))) return rc;
    DBG("stage: team tables");
    // ---- team kernels (tf_eri_team.hip.h): the uncontracted classes of the per-class mode, packed layout.  Per pair class the ket
    // pair transform (Cartesian component pairs -> output pairs inside each x/y parity class, normalisation ratios folded in), per
    // shell pair the slab offsets of its output pairs.
    if (use_team) {
        std::vector<int> kt_ptr, kt_k, kq_ptr(npairs, 0), kq_off;
        std::vector<double> kt_c, blkC, blkD;
        std::vector<std::vector<int>> kap_list(ncls);               // output pairs (sc << 8 | sd) of a pair class, sorted by parity class
        auto rows_of = [&](const tf::Shell &sh, std::vector<double> &U) {   // transformation rows of a shell (identity: Cartesian output)
            if (spherical) { tf::sph_block(sh.L, U); return; }
            U.assign((size_t)sh.ncomp * sh.ncomp, 0.0);
            for (int i = 0; i < sh.ncomp; ++i) U[(size_t)i * sh.ncomp + i] = 1.0;
        };
        auto cls_of = [&](const tf::Shell &sh, const std::vector<double> &U, int r) {   // parity class of an output function: its first component's
            int cc = 0;
            while (cc + 1 < sh.ncomp && U[(size_t)r * sh.ncomp + cc] == 0.0) ++cc;
            const int a = sh.comp_off + cc;
            return (bs.c_lx[a] & 1) | ((bs.c_ly[a] & 1) << 1);
        };
        for (int c = 0; c < ncls; ++c) {
            const DPair &pr = ctx->host_pairs[ctx->class_pairs[c][0]];
            const tf::Shell &sC = bs.shells[pr.A], &sD = bs.shells[pr.B];
            const int nsc = out_dim(sC), nsd = out_dim(sD), ncc = sC.ncomp, ncd = sD.ncomp;
            rows_of(sC, blkC); rows_of(sD, blkD);
            std::vector<int> loc((size_t)ncc * ncd, 0);              // position of a component pair inside its parity class (inverse of ct_ord)
            for (int cl = 0; cl < 4; ++cl)
                for (int s2 = pr.pcls[cl]; s2 < pr.pcls[cl + 1]; ++s2) loc[ctx->h_ct_ord[pr.tab_off + s2]] = s2 - pr.pcls[cl];
            KClassTab &kt = kct[c];
            kt.tp_off = (int)kt_ptr.size(); kt.te_off = (int)kt_k.size();
            for (int cl = 0; cl < 4; ++cl) {
                kt.pS[cl] = (int)kap_list[c].size();
                for (int sc = 0; sc < nsc; ++sc)
                    for (int sd = 0; sd < nsd; ++sd) {
                        if ((cls_of(sC, blkC, sc) ^ cls_of(sD, blkD, sd)) != cl) continue;
                        kap_list[c].push_back((sc << 8) | sd);
                        kt_ptr.push_back((int)kt_k.size() - kt.te_off);
                        for (int cc = 0; cc < ncc; ++cc) {
                            const double uc = blkC[(size_t)sc * ncc + cc];
                            if (uc == 0.0) continue;
                            for (int cd2 = 0; cd2 < ncd; ++cd2) {
                                const double ud = blkD[(size_t)sd * ncd + cd2];
                                if (ud == 0.0) continue;
                                const int f = cc * ncd + cd2;
                                kt_k.push_back(loc[f]);
                                kt_c.push_back(uc * ud * ctx->h_ct_sc[pr.tab_off + f]);
                            }
                        }
                    }
            }
            kt.pS[4] = kt.nkap = (int)kap_list[c].size();
            kt_ptr.push_back((int)kt_k.size() - kt.te_off);
            kt.nnzT = (int)kt_k.size() - kt.te_off;
        }
        for (int p = 0; p < npairs; ++p) {
            const DPair &pr = ctx->host_pairs[p];
            kq_ptr[p] = (int)kq_off.size();
            for (int word : kap_list[ctx->pair_class[p]]) {
                const int k = pr.outoff_a + (word >> 8), l = pr.outoff_b + (word & 255);
                if (l > k) { kq_off.push_back(-1); continue; }
                const int cb = H.cls[k] ^ H.cls[l];
                kq_off.push_back(H.fullsec[cb][H.cls[k]] + H.kinfo[(size_t)cb * N + H.sigma[k]].offA + H.loc[l]);
            }
        }
        if ((rc = upload(ctx, kq_ptr, &d_kq_ptr, false)) || (rc = upload(ctx, kq_off, &d_kq_off, false)) || (rc = upload(ctx, kt_ptr, &d_kt_ptr, false)) ||
            (rc = upload(ctx, kt_k, &d_kt_k, false)) || (rc = upload(ctx, kt_c, &d_kt_c, false)))
            return rc;
        ctx->db.kq_ptr = d_kq_ptr; ctx->db.kq_off = d_kq_off; ctx->db.kt_ptr = d_kt_ptr; ctx->db.kt_k = d_kt_k; ctx->db.kt_c = d_kt_c;
        // flat records of the class-sorted ket list (uncontracted pairs) and, per class and shell A, the kets with first shell <= A
        std::vector<KetRec> krec(ket_sorted.size());
        for (size_t k = 0; k < ket_sorted.size(); ++k) {
            const tf::Pair &pr = bs.pairs[ket_sorted[k]];
            const double qq = bs.pp_p[pr.pp_off];
            krec[k] = KetRec{qq, bs.pp_Pz[pr.pp_off], bs.pp_K[pr.pp_off] / qq, (unsigned)pr.e_off, kq_ptr[ket_sorted[k]]};
        }
        std::vector<int> kcnt((size_t)ncls * nsh, 0);
        for (int c = 0; c < ncls; ++c) {
            for (int p2 : ctx->class_pairs[c]) ++kcnt[(size_t)c * nsh + bs.pairs[p2].A];
            for (int a = 1; a < nsh; ++a) kcnt[(size_t)c * nsh + a] += kcnt[(size_t)c * nsh + a - 1];
        }
        if ((rc = upload(ctx, krec, &d_ketrec, false)) || (rc = upload(ctx, kcnt, &d_kcnt, false))) return rc;
        if (use_team_pc) {
            // flat lists of the parity-allowed components and of the outputs of every (bra class, ket class) of uncontracted pairs
            std::vector<int> tflat;
            flat_off.assign((size_t)ncls * ncls, -1);
            for (int bc = 0; bc < ncls; ++bc)
                for (int kc = 0; kc < ncls; ++kc) {
                    const DPair &hb = ctx->host_pairs[ctx->class_pairs[bc][0]], &hk = ctx->host_pairs[ctx->class_pairs[kc][0]];
                    if (hb.npp != 1 || hk.npp != 1 || hb.La + hb.Lb > TF_TEAM_LMAX || hk.La + hk.Lb > TF_TEAM_LMAX) continue;
                    int mb, mk, nz;
                    const TClass t = tclass_common(bc, kc, mb, mk, nz);
                    if (nz > 2048) continue;                          // (chunked classes keep the loop per parity class)
                    flat_off[(size_t)bc * ncls + kc] = (int)tflat.size();
                    for (int c = 0; c < 4; ++c)
                        for (int il = 0; il < t.pA[c + 1] - t.pA[c]; ++il)
                            for (int kl = 0; kl < t.pK[c + 1] - t.pK[c]; ++kl) tflat.push_back((t.pA[c] + il) | ((t.pK[c] + kl) << 16));
                    if (tflat.size() & 1) tflat.push_back(0);
                    int vbase = 0;
                    for (int c = 0; c < 4; ++c) {
                        const int nA = t.pA[c + 1] - t.pA[c], nK = t.pK[c + 1] - t.pK[c], nS = t.pS[c + 1] - t.pS[c];
                        for (int il = 0; il < nA; ++il)
                            for (int ks = 0; ks < nS; ++ks) { tflat.push_back((t.pA[c] + il) | ((t.pS[c] + ks) << 16)); tflat.push_back(vbase + il * nK); }
                        vbase += nA * nK;
                    }
                }
            if ((rc = upload(ctx, tflat, &d_tflat, false))) return rc;
            ctx->db.tflat = d_tflat;
        }
    }
    if (!per_class)
        std::stable_sort(mine_sorted.begin(), mine_sorted.end(), [&](int x, int y) {
            const int gx = pair_group(x), gy = pair_group(y);
            return gx != gy ? gx < gy : pair_cost(x) > pair_cost(y);
        });

    // device index buffers sized for the largest possible slab, reused by every slab
    size_t max_out = 0;
    for (int p : mine_sorted) max_out = std::max<size_t>(max_out, (size_t)pair_rows[p]);
    const size_t cap_bra = std::min<size_t>(mine_sorted.size(), 65535) + 1;
    const size_t cap_out = (size_t)std::min<long long>(ctx->n_rows, max_rows_c * 2 + (long long)max_out) + 1;
    int *d_bra = nullptr; long long *d_braoff = nullptr; void *d_out = nullptr;
    signed char *d_rowcls = nullptr;                               // parity class of every slab row (small-problem mode, packed layout)
    // two sets of a slab's index lists: the lists of slab k + 1 are uploaded (on a stream of their own) while the kernels of slab k run;
    // the kernels themselves stay ordered by the streams (one half-transformed slab)
    const size_t out_bytes = cap_out * std::max(std::max(sizeof(OutRow), sizeof(OutRowP)), sizeof(OutRowT)), rowcls_bytes = (size_t)max_rows_c + 1;
    HIPCHK(ctx, tf_malloc((void **)&d_bra, 2 * cap_bra * sizeof(int)));
    HIPCHK(ctx, tf_malloc((void **)&d_braoff, 2 * cap_bra * sizeof(long long)));
    if (use_team_pc) HIPCHK(ctx, tf_malloc((void **)&d_brarec, 2 * cap_bra * sizeof(BraRec)));
    HIPCHK(ctx, tf_malloc((void **)&d_out, 2 * out_bytes));
    if (packed && !per_class) HIPCHK(ctx, tf_malloc((void **)&d_rowcls, 2 * rowcls_bytes));
    int *const d_bra_alloc = d_bra; long long *const d_braoff_alloc = d_braoff; void *const d_out_alloc = d_out;
    BraRec *const d_brarec_alloc = d_brarec; signed char *const d_rowcls_alloc = d_rowcls;
    d_bra_base = d_bra;
    long long slab_no = 0;
    std::vector<hipEvent_t> tev;                                   // 4 timing events per slab, read at the end
    std::vector<hipEvent_t> tev2;                                  // 2 per slab around the task-list team kernels (they count as ERI kernels)
    DBG("stage: teamc tables");
    // ---- small-problem mode with team kernels (eri_teamc_kernel): class records per (bra class, ket class), created on demand; tasks
    // per slab.  A quartet belongs to that kernel when both pair sums are <= TF_TEAM_LMAX and it has <= teamc_pqmax primitive quartets;
    // eri_cfact_kernel skips exactly those (CFCaps::team_lmax / team_pqmax).
    std::vector<TClass> tcs_host;
    std::vector<int> tc_of((size_t)ncls * ncls, -2), tc_team;                // -2: not made yet, -1: not eligible
    std::vector<size_t> tc_lds;
    auto teamc_class = [&](int bcls, int kcls) -> int {
        int &slot = tc_of[(size_t)bcls * ncls + kcls];
        if (slot != -2) return slot;
        int maxblk, maxK, nnzc;
        TClass t = tclass_common(bcls, kcls, maxblk, maxK, nnzc);
        const int LAB = t.La + t.Lb, LCD = t.Lc + t.Ld, L = LAB + LCD, NM = L / 2 + 1, XS = NM | 1;
        if (LAB > TF_TEAM_LMAX || LCD > TF_TEAM_LMAX) return slot = -1;
        auto even = [](int x) { return (x + 1) & ~1; };
        int o = 0;
        t.oE12 = 0;
        t.oOffA = o; o += 2 * t.nab;
        t.oScA = o; o += even(t.nab);
        t.oOffK = o; o += 2 * t.ncd;
        t.oTp = o; o += even((t.nkap + 2) / 2);
        t.oTk = o; o += even((t.nnzT + 1) / 2);
        t.oTc = o; o += even(t.nnzT);
        t.shared_doubles = o;
        t.vcap = even(2 * t.nEab + 2 * t.nEcd + (L + 1) * (L + 2) + t.nTcd * (LAB + 1) * NM);      // E12, E34, R, G of one primitive quartet
        t.team_doubles = 2 * t.nT * XS + t.vcap + even(t.nacc) + even((t.nkap + 1) / 2);
        auto bytes_of = [&](int tm) { return ((size_t)t.shared_doubles + (size_t)(256 / tm) * t.team_doubles) * sizeof(double); };
        int team = 0;
        if (t.nT <= 16 && nnzc <= 96 && eri_team_available(LAB, LCD, 16)) team = 16;
        else if (eri_team_available(LAB, LCD, 64) && bytes_of(64) <= 52 * 1024) team = 64;
        else if (eri_team_available(LAB, LCD, 256) && bytes_of(256) <= 160 * 1024 - 256) team = 256;
        else if (eri_team_available(LAB, LCD, 64) && bytes_of(64) <= 160 * 1024 - 256) team = 64;
        if (!team) return slot = -1;
        tcs_host.push_back(t); tc_team.push_back(team); tc_lds.push_back(bytes_of(team));
        return slot = (int)tcs_host.size() - 1;
    };
    bool any_wide_pair = false;                                    // a pair sum beyond the team kernels' instantiations
    int max_npp_all = 1;
    for (int p2 = 0; p2 < npairs; ++p2) {
        any_wide_pair = any_wide_pair || bs.pairs[p2].La + bs.pairs[p2].Lb > TF_TEAM_LMAX;
        max_npp_all = std::max(max_npp_all, bs.pairs[p2].npp);
    }
    TClass *d_tcs = nullptr; TeamTask *d_tasks = nullptr;
    size_t d_tcs_cap = 0, d_tasks_cap = 0;
    size_t cursor = 0;
    DBG("stage: slab loop");
    while (cursor < mine_sorted.size()) {
        std::vector<int> bra; std::vector<long long> braoff; std::vector<OutRow> outs;
        std::vector<OutRowP> outsP;
        std::vector<OutRowT> outsT;
        std::vector<signed char> rowcls;
        long long rows_c = 0;
        while (cursor < mine_sorted.size()) {
            const int p = mine_sorted[cursor];
            const tf::Shell &a = bs.shells[bs.pairs[p].A], &b = bs.shells[bs.pairs[p].B];
            const long long nr = (long long)a.ncomp * b.ncomp;
            if (!bra.empty() && (rows_c + nr > max_rows_c || bra.size() >= 65535 || outs.size() + outsP.size() + outsT.size() + (size_t)pair_rows[p] >= cap_out)) break;
            bra.push_back(p); braoff.push_back(rows_c);
            long long r = pair_first_row[p];
            for (int x = 0; x < out_dim(a); ++x)
                for (int y = 0; y < out_dim(b); ++y) {
                    const int i = out_off(a) + x, j = out_off(b) + y;
                    if (i < j) continue;
                    if (!packed) { outs.push_back(OutRow{i, j, a.cart_off, b.cart_off, b.ncomp, 0, rows_c, r++}); continue; }
                    if (tiles) { outsT.push_back(OutRowT{i, j, H.sigma[i], H.sigma[j], H.cls[i] ^ H.cls[j], b.ncomp, a.cart_off, b.cart_off, rows_c}); continue; }
                    const int lr = rowmap[ikey(H.sigma[i], H.sigma[j])];
                    OutRowP o{};
                    o.i = i; o.j = j; o.iI = H.sigma[i]; o.lamj = H.loc[j]; o.c = H.cls[i] ^ H.cls[j]; o.ncb = b.ncomp;
                    o.cartA = a.cart_off; o.cartB = b.cart_off;
                    for (int t = 0; t < 4; ++t) o.secoff[t] = rowsec[6 * (size_t)lr + t];
                    o.len = rowlen[lr]; o.slab_off = rows_c; o.ubase = rowoff[lr]; o.upos = rowsec[6 * (size_t)lr + 4]; o.unr = rowsec[6 * (size_t)lr + 5];
                    outsP.push_back(o);
                }
            if (packed && !per_class)
                for (int ca = 0; ca < a.ncomp; ++ca)
                    for (int cb = 0; cb < b.ncomp; ++cb) {
                        const int u = a.comp_off + ca, v = b.comp_off + cb;
                        rowcls.push_back((signed char)(((bs.c_lx[u] + bs.c_lx[v]) & 1) | (((bs.c_ly[u] + bs.c_ly[v]) & 1) << 1)));
                    }
            rows_c += nr;
            const int Alim = packed ? bs.pairs[p].A + 1 : nsh;
            n_quart += cum_pairs[Alim];
            n_primq += (long long)bs.pairs[p].npp * cum_pp[Alim];
            n_compq += nr * cum_comp[Alim];
            {
                const double npp_ab = (double)bs.pairs[p].npp;
                const int lab = bs.pairs[p].La + bs.pairs[p].Lb;
                for (int c = 0; c < 4; ++c) {
                    nominal_flops += 8.0 * npp_ab * pairW[p][c] * cumW[Alim][c];
                    for (int lcd = 0; lcd <= 10; ++lcd) {
                        const double L1 = (double)(lab + lcd + 1);
                        nominal_flops += npp_ab * pairNc[p][c] * cumNL[Alim][4 * lcd + c] * (6.0 * L1 + 1.5 * L1 * L1 + 60.0);
                    }
                }
            }
            ++cursor;
        }
        // this set of index lists was last read by the kernels of the slab before the previous one: wait for that slab only
        const int set = (int)(slab_no & 1);
        if (slab_no >= 2) HIPCHK(ctx, hipEventSynchronize(ctx->slab_done[set]));
        d_bra = d_bra_alloc + (size_t)set * cap_bra; d_braoff = d_braoff_alloc + (size_t)set * cap_bra;
        d_out = (char *)d_out_alloc + (size_t)set * out_bytes;
        if (d_brarec_alloc) d_brarec = d_brarec_alloc + (size_t)set * cap_bra;
        if (d_rowcls_alloc) d_rowcls = d_rowcls_alloc + (size_t)set * rowcls_bytes;
        d_bra_base = d_bra;
        ++slab_no;
        const hipStream_t cst = ctx->cstream;
        HIPCHK(ctx, hipMemcpyAsync(d_bra, bra.data(), bra.size() * sizeof(int), hipMemcpyHostToDevice, cst));
        HIPCHK(ctx, hipMemcpyAsync(d_braoff, braoff.data(), braoff.size() * sizeof(long long), hipMemcpyHostToDevice, cst));
        std::vector<BraRec> brec;
        if (use_team_pc) {
            brec.resize(bra.size());
            for (size_t k = 0; k < bra.size(); ++k) {
                const tf::Pair &pr = bs.pairs[bra[k]];
                const double pp = bs.pp_p[pr.pp_off];
                brec[k] = BraRec{pp, bs.pp_Pz[pr.pp_off], bs.pp_K[pr.pp_off] / pp, (unsigned)pr.e_off, pr.A, braoff[k], 0};
            }
            HIPCHK(ctx, hipMemcpyAsync(d_brarec, brec.data(), brec.size() * sizeof(BraRec), hipMemcpyHostToDevice, cst));
        }
        if (!outs.empty()) HIPCHK(ctx, hipMemcpyAsync(d_out, outs.data(), outs.size() * sizeof(OutRow), hipMemcpyHostToDevice, cst));
        if (!outsP.empty()) HIPCHK(ctx, hipMemcpyAsync(d_out, outsP.data(), outsP.size() * sizeof(OutRowP), hipMemcpyHostToDevice, cst));
        if (!outsT.empty()) HIPCHK(ctx, hipMemcpyAsync(d_out, outsT.data(), outsT.size() * sizeof(OutRowT), hipMemcpyHostToDevice, cst));
        if (!rowcls.empty()) HIPCHK(ctx, hipMemcpyAsync(d_rowcls, rowcls.data(), rowcls.size(), hipMemcpyHostToDevice, cst));
        HIPCHK(ctx, hipStreamSynchronize(cst));                // (pageable sources: the copies are complete; the host vectors may go)
        DBG("slab: %zu bra pairs, %lld cart rows, %zu out rows", bra.size(), rows_c, outs.size() + outsP.size() + outsT.size());
        hipEvent_t e4[4];
        for (auto &e : e4) { HIPCHK(ctx, hipEventCreate(&e)); tev.push_back(e); }
        if (per_class && !packed && ld != N) HIPCHK(ctx, hipMemsetAsync(d_T2, 0, (size_t)rows_c * N * ld * sizeof(double), 0));   // pad columns
        // small-problem mode with team kernels: the slab's tasks, one launch per (LAB, LCD, team, LDS size class)
        struct TcLaunch { int LAB, LCD, team; size_t lds; std::vector<TeamTask> tasks; std::vector<double> cost; };
        std::vector<TcLaunch> tcl;
        bool old_needed = !use_teamc;
        if (use_teamc) {
            std::map<std::array<int, 4>, int> lidx;
            int max_npp_bra = 1;
            for (size_t b = 0; b < bra.size(); ++b) {
                const int pb = bra[b], A = bs.pairs[pb].A, bcls = ctx->pair_class[pb];
                max_npp_bra = std::max(max_npp_bra, bs.pairs[pb].npp);
                for (int kc = 0; kc < ncls; ++kc) {
                    const std::vector<int> &kl = ctx->class_pairs[kc];
                    const int nk_all = (int)(std::upper_bound(kl.begin(), kl.end(), A, [&](int a, int p2) { return a < bs.pairs[p2].A; }) - kl.begin());
                    if (nk_all == 0) continue;
                    const int id = teamc_class(bcls, kc);
                    if (id < 0) continue;
                    const int team = tc_team[(size_t)id], NT = 256 / team;
                    int lg = 10;
                    while (((size_t)1 << lg) < tc_lds[(size_t)id]) ++lg;
                    const std::array<int, 4> key{tcs_host[(size_t)id].La + tcs_host[(size_t)id].Lb, tcs_host[(size_t)id].Lc + tcs_host[(size_t)id].Ld, team, lg};
                    auto it = lidx.find(key);
                    if (it == lidx.end()) { it = lidx.emplace(key, (int)tcl.size()).first; tcl.push_back(TcLaunch{key[0], key[1], team, 0, {}, {}}); }
                    TcLaunch &TL = tcl[(size_t)it->second];
                    TL.lds = std::max(TL.lds, tc_lds[(size_t)id]);
                    for (int g0 = 0; g0 < nk_all; g0 += NT) {
                        const int nk = std::min(NT, nk_all - g0);
                        int npq_max = 0, todo = 0;
                        for (int u = 0; u < nk; ++u) {
                            const int npq = bs.pairs[pb].npp * bs.pairs[kl[(size_t)(g0 + u)]].npp;
                            if (npq <= teamc_pqmax) { ++todo; npq_max = std::max(npq_max, npq); }
                        }
                        if (!todo) continue;                       // (every quartet of the group goes to eri_cfact_kernel)
                        TL.tasks.push_back(TeamTask{id, pb, ket_off[kc] + g0, nk, braoff[b]});
                        TL.cost.push_back((double)npq_max * (tcs_host[(size_t)id].nT + tcs_host[(size_t)id].nacc));
                    }
                }
            }
            old_needed = any_wide_pair || (long long)max_npp_bra * max_npp_all > teamc_pqmax;
            // heaviest tasks first inside a launch (workgroups are dispatched in index order), all tasks in one device array
            std::vector<TeamTask> all;
            for (TcLaunch &TL : tcl) {
                std::vector<int> ord(TL.tasks.size());
                std::iota(ord.begin(), ord.end(), 0);
                std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return TL.cost[(size_t)x] > TL.cost[(size_t)y]; });
                std::vector<TeamTask> sorted(TL.tasks.size());
                for (size_t k = 0; k < ord.size(); ++k) sorted[k] = TL.tasks[(size_t)ord[k]];
                TL.tasks.swap(sorted);
                all.insert(all.end(), TL.tasks.begin(), TL.tasks.end());
            }
            if (tcs_host.size() > d_tcs_cap) {
                if (d_tcs) (void)tf_free(d_tcs);
                d_tcs_cap = tcs_host.size() + 64;
                HIPCHK(ctx, tf_malloc((void **)&d_tcs, d_tcs_cap * sizeof(TClass)));
            }
            if (all.size() > d_tasks_cap) {
                if (d_tasks) (void)tf_free(d_tasks);
                d_tasks_cap = all.size() + 1024;
                HIPCHK(ctx, tf_malloc((void **)&d_tasks, d_tasks_cap * sizeof(TeamTask)));
            }
            if (!tcs_host.empty()) HIPCHK(ctx, hipMemcpy(d_tcs, tcs_host.data(), tcs_host.size() * sizeof(TClass), hipMemcpyHostToDevice));
            if (!all.empty()) HIPCHK(ctx, hipMemcpy(d_tasks, all.data(), all.size() * sizeof(TeamTask), hipMemcpyHostToDevice));
        }
        // generic mode: eri_cfact_kernel stores only the components that are not zero by x/y parity
        if (!per_class && old_needed) HIPCHK(ctx, hipMemsetAsync(d_C, 0, (size_t)rows_c * Nc * Nc * sizeof(double), 0));
        HIPCHK(ctx, hipEventRecord(e4[0], 0));
        for (int k = 0; k < NSTREAM; ++k) HIPCHK(ctx, hipStreamWaitEvent(streams[k], e4[0], 0));
        // runs of equal bra class inside the slab
        size_t r0 = 0;
        if (!per_class) { if (old_needed && (rc = generic_launch(bra, d_bra, d_braoff))) return rc; r0 = bra.size(); }
        while (r0 < bra.size()) {
            const int bcls = ctx->pair_class[bra[r0]];
            size_t r1 = r0;
            int max_npp = 1, Amax = 0;
            while (r1 < bra.size() && ctx->pair_class[bra[r1]] == bcls) {
                max_npp = std::max(max_npp, bs.pairs[bra[r1]].npp);
                Amax = std::max(Amax, bs.pairs[bra[r1]].A);
                ++r1;
            }
            for (int kcls = 0; kcls < ncls; ++kcls)
                class_launch(bcls, kcls, max_npp, (unsigned)(r1 - r0), d_bra + r0, d_braoff + r0, Amax);
            r0 = r1;
        }
        for (int k = 0; k < NSTREAM; ++k) {
            HIPCHK(ctx, hipEventRecord(sev[k], streams[k]));
            HIPCHK(ctx, hipStreamWaitEvent(0, sev[k], 0));
        }
        HIPCHK(ctx, hipEventRecord(e4[1], 0));
        if (!per_class && old_needed) {
            for (long long r0s = 0; r0s < rows_c; r0s += 65535) {      // both ket axes in one pass over the slab
                const unsigned ny = (unsigned)std::min<long long>(65535, rows_c - r0s);
                if (packed)
                    hipLaunchKernelGGL(xform_ket_packed, dim3((unsigned)((N + 3) / 4), ny), dim3(256), 0, 0, d_C + (size_t)r0s * Nc * Nc,
                                       d_T2 + (size_t)r0s * H.RLS, Nc, ctx->bl, (long long)H.RLS, d_rowcls + r0s, ctx->d_csr_ptr, ctx->d_csr_idx,
                                       ctx->d_csr_val);
                else
                    hipLaunchKernelGGL(xform_ket_both, dim3((unsigned)((N + 3) / 4), ny), dim3(256), 0, 0, d_C + (size_t)r0s * Nc * Nc,
                                       d_T2 + (size_t)r0s * N * ld, Nc, N, ld, ctx->d_csr_ptr, ctx->d_csr_idx, ctx->d_csr_val, 0);
            }
        }
        if (use_teamc && !tcl.empty()) {
            // the team kernels write their quartets' part of the half-transformed slab (behind the ket transform of the others' part,
            // which has left zeros there); heaviest launches first, spread over the streams
            std::vector<size_t> first(tcl.size(), 0), order(tcl.size());
            for (size_t k = 1; k < tcl.size(); ++k) first[k] = first[k - 1] + tcl[k - 1].tasks.size();
            std::iota(order.begin(), order.end(), 0);
            auto total = [&](size_t k) { double c = 0; for (double x : tcl[k].cost) c += x; return c; };
            std::vector<double> tot(tcl.size());
            for (size_t k = 0; k < tcl.size(); ++k) tot[k] = total(k);
            std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return tot[x] > tot[y]; });
            hipEvent_t et[2];
            for (auto &e : et) { HIPCHK(ctx, hipEventCreate(&e)); tev2.push_back(e); }
            HIPCHK(ctx, hipEventRecord(et[0], 0));
            HIPCHK(ctx, hipEventRecord(sev[0], 0));
            for (int k = 0; k < NSTREAM; ++k) HIPCHK(ctx, hipStreamWaitEvent(streams[k], sev[0], 0));
            int nl = 0;
            for (size_t k : order) {
                const TcLaunch &TL = tcl[k];
                if (TL.tasks.empty()) continue;
                TeamcLaunch a{TL.LAB, TL.LCD, TL.team, (unsigned)TL.tasks.size(), TL.lds, streams[nl++ % NSTREAM], &ctx->db, d_tcs, d_tasks + first[k], d_kets,
                              teamc_pqmax, d_T2};
                const hipError_t e = eri_teamc_launch(a);
                if (e != hipSuccess && team_error == hipSuccess) team_error = e;
                ++launch_count;
            }
            for (int k = 0; k < NSTREAM; ++k) {
                HIPCHK(ctx, hipEventRecord(sev[k], streams[k]));
                HIPCHK(ctx, hipStreamWaitEvent(0, sev[k], 0));
            }
            HIPCHK(ctx, hipEventRecord(et[1], 0));
        }
        HIPCHK(ctx, hipEventRecord(e4[2], 0));
        if (!outs.empty()) {
            const unsigned gx = (unsigned)std::min<long long>((row_len + 255) / 256, 4096);
            for (size_t o0 = 0; o0 < outs.size(); o0 += 65535) {
                const unsigned ny = (unsigned)std::min<size_t>(65535, outs.size() - o0);
                hipLaunchKernelGGL(xform_bra_store, dim3(gx, ny), dim3(256), 0, 0, d_T2, ctx->d_eri, reinterpret_cast<const OutRow *>(d_out) + o0,
                                   row_len, ctx->d_csr_ptr, ctx->d_csr_idx, ctx->d_csr_val);
            }
        }
        if (!outsP.empty()) {
            int maxlen = 1;
            for (const OutRowP &o : outsP) maxlen = std::max(maxlen, o.len);
            for (size_t o0 = 0; o0 < outsP.size(); o0 += 65535) {
                const unsigned ny = (unsigned)std::min<size_t>(65535, outsP.size() - o0);
                hipLaunchKernelGGL(xform_bra_store_packed, dim3((unsigned)((maxlen + 255) / 256), ny), dim3(256), 0, 0, d_T2, ctx->d_eri,
                                   reinterpret_cast<const OutRowP *>(d_out) + o0, (long long)H.RLS, ctx->bl, ctx->d_csr_ptr, ctx->d_csr_idx,
                                   ctx->d_csr_val);
            }
        }
        if (!outsT.empty()) {
            for (size_t o0 = 0; o0 < outsT.size(); o0 += 65535) {
                const unsigned ny = (unsigned)std::min<size_t>(65535, outsT.size() - o0);
                hipLaunchKernelGGL(xform_bra_store_tiles, dim3((unsigned)((H.RLS + 255) / 256), ny), dim3(256), 0, 0, d_T2, ctx->d_eri,
                                   reinterpret_cast<const OutRowT *>(d_out) + o0, (long long)H.RLS, ctx->bl, ctx->tv, ctx->d_csr_ptr, ctx->d_csr_idx,
                                   ctx->d_csr_val);
            }
        }
        HIPCHK(ctx, hipEventRecord(e4[3], 0));
        HIPCHK(ctx, hipEventRecord(ctx->slab_done[set], 0));
    }
    DBG("all slabs launched (%d class launches)", launch_count);
    if ((rc = consumer_tables())) { (void)hipDeviceSynchronize(); return rc; }
    DBG("consumer tables built");
    HIPCHK(ctx, hipDeviceSynchronize());
    DBG("device drained");
    HIPCHK(ctx, hipGetLastError());
    for (size_t k = 0; k + 3 < tev.size(); k += 4) {
        t_stage[1] += seconds_between(tev[k], tev[k + 1]);
        t_stage[2] += seconds_between(tev[k + 1], tev[k + 2]);
        t_stage[3] += seconds_between(tev[k + 2], tev[k + 3]);
    }
    for (size_t k = 0; k + 1 < tev2.size(); k += 2) {
        const double dt = seconds_between(tev2[k], tev2[k + 1]);
        t_stage[1] += dt; t_stage[2] -= dt;
    }
    for (hipEvent_t e : tev) (void)hipEventDestroy(e);
    for (hipEvent_t e : tev2) (void)hipEventDestroy(e);
    (void)tf_free(d_bra_alloc); (void)tf_free(d_braoff_alloc); (void)tf_free(d_out_alloc);
    if (d_rowcls_alloc) (void)tf_free(d_rowcls_alloc);
    d_brarec = d_brarec_alloc;
    (void)tf_free(d_kets); (void)tf_free(d_kets_all);
    for (void *pt : {(void *)d_fam_heads, (void *)d_fam_ptr, (void *)d_fam_mem})
        if (pt) (void)tf_free(pt);
    for (void *pt : fam_allocs) (void)tf_free(pt);
    for (void *pt : {(void *)d_kq_ptr, (void *)d_kq_off, (void *)d_kt_ptr, (void *)d_kt_k, (void *)d_kt_c, (void *)d_ketrec, (void *)d_brarec, (void *)d_kcnt,
                     (void *)d_tcs, (void *)d_tasks, (void *)d_tflat})
        if (pt) (void)tf_free(pt);
    ctx->db.kq_ptr = ctx->db.kq_off = ctx->db.kt_ptr = ctx->db.kt_k = nullptr; ctx->db.kt_c = nullptr;
    if (team_error != hipSuccess) TF_FAIL(ctx, TF_ENODEVICE, "launch of a team ERI kernel failed: %s", hipGetErrorString(team_error));
    t_stage[0] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_wall0).count();
    std::copy(t_stage, t_stage + 4, ctx->eri_seconds);
    ctx->eri_counts[0] = n_quart; ctx->eri_counts[1] = n_primq; ctx->eri_counts[2] = n_compq;
    ctx->eri_nominal_flops = nominal_flops;

    DBG("scratch freed");
    // ---- J/K scratch
    const size_t nn = (size_t)N * N;
    // (sized for two densities per pass)
    const int NWjk = packed ? std::max(1, H.NW) * H.MP : 1;         // column chunks of jk_packed_kernel x parts of a walk
    ctx->cd_bytes[0] = 2 * (size_t)std::max(1, NWjk) * std::max<size_t>(1, (size_t)ctx->n_rows) * sizeof(double);
    HIPCHK(ctx, tf_malloc((void **)&ctx->d_Jrow, 2 * (size_t)std::max(1, NWjk) * std::max<size_t>(1, (size_t)ctx->n_rows) * sizeof(double)));
    HIPCHK(ctx, hipMemset(ctx->d_Jrow, 0, 2 * (size_t)std::max(1, NWjk) * std::max<size_t>(1, (size_t)ctx->n_rows) * sizeof(double)));
    if (tiles) {
        // partial sums of jk_tile_kernel (one density per pass for now) and the per-pass outputs of its reductions
        const tf_ctx::TileList &D = ctx->tl1;
        const size_t pml = (size_t)std::max(1, ctx->tiles.pm_len);
        auto alloc_t = [&](double **p, size_t doubles, bool zero) -> int {
            HIPCHK(ctx, tf_malloc((void **)p, std::max<size_t>(1, doubles) * sizeof(double)));
            ctx->tile_allocs.push_back(*p);
            if (zero) HIPCHK(ctx, hipMemset(*p, 0, std::max<size_t>(1, doubles) * sizeof(double)));
            return TF_OK;
        };
        if ((rc = alloc_t(&ctx->t_X, 8 * nn, false)) || (rc = alloc_t(&ctx->t_Pm, 8 * pml, true)) || (rc = alloc_t(&ctx->t_out, 2 * 6 * 8 * nn, true)) ||
            (rc = alloc_t(&ctx->t_JtTot, 8 * pml, true)))
            return rc;
        (void)D;
    } else if (packed) {
        // everything sized for a two-density pass (second density behind the first)
        const size_t npr = (size_t)std::max<long long>(1, H.NPtot);    // padded pair index space
        const size_t ny = (size_t)std::max(ctx->jkt[0].ypart_len, 2 * ctx->jkt[1].ypart_len);
        const size_t ng = (size_t)std::max(ctx->jkt[0].n_groups, 2 * ctx->jkt[1].n_groups);
        const int nsegmax = std::max(ctx->jkt[0].nseg, ctx->jkt[1].nseg);
        HIPCHK(ctx, tf_malloc((void **)&ctx->d_Psym, 2 * nn * sizeof(double)));
        HIPCHK(ctx, tf_malloc((void **)&ctx->d_Pp, 2 * npr * sizeof(double)));
        HIPCHK(ctx, hipMemset(ctx->d_Pp, 0, 2 * npr * sizeof(double)));   // pad slots stay zero
        HIPCHK(ctx, tf_malloc((void **)&ctx->d_ypart, std::max<size_t>(1, ny) * sizeof(double)));
        // column parts [.][N] followed by the row parts [.][RS]
        // partial sums: [one-density pass | two-density pass], each [column parts | row parts] per density.  The passes have separate
        // regions (their groups differ), every region is zeroed once: the set of entries a pass writes does not depend on the density,
        // and the reductions rely on the entries no task writes being zero.
        const size_t per_g = (size_t)N * H.MP + H.RS, nr1 = std::max<size_t>(1, (size_t)ctx->n_rows);
        const size_t di_doubles = ((size_t)std::max(1, ctx->jkt[0].n_groups) + 2 * (size_t)std::max(1, ctx->jkt[1].n_groups)) * per_g;
        const size_t dj_doubles = 3 * nr1 * per_g;
        ctx->cd_bytes[1] = std::max<size_t>(1, (size_t)ctx->jkt[0].ypart_len + 2 * (size_t)ctx->jkt[1].ypart_len) * sizeof(double); ctx->cd_bytes[2] = di_doubles * sizeof(double); ctx->cd_bytes[3] = dj_doubles * sizeof(double);
        HIPCHK(ctx, tf_malloc((void **)&ctx->d_DI, di_doubles * sizeof(double)));
        HIPCHK(ctx, tf_malloc((void **)&ctx->d_DJ, dj_doubles * sizeof(double)));
        HIPCHK(ctx, hipMemset(ctx->d_DI, 0, di_doubles * sizeof(double)));
        HIPCHK(ctx, hipMemset(ctx->d_DJ, 0, dj_doubles * sizeof(double)));
        HIPCHK(ctx, hipMemset(ctx->d_ypart, 0, std::max<size_t>(1, ny) * sizeof(double)));
        HIPCHK(ctx, tf_malloc((void **)&ctx->d_Jt, 2 * (size_t)nsegmax * npr * sizeof(double)));
        HIPCHK(ctx, tf_malloc((void **)&ctx->d_D, 2 * nn * sizeof(double)));
    } else
        HIPCHK(ctx, tf_malloc((void **)&ctx->d_Kp, 2 * std::max<size_t>(1, (size_t)ctx->n_rows) * 2 * ld * sizeof(double)));
    HIPCHK(ctx, tf_malloc((void **)&ctx->d_Ppad, 2 * (size_t)N * ld * sizeof(double)));
    const size_t npass = tiles ? 8 : 2;                             // densities of one call of tf_fock_jk that go through the tensor together
    HIPCHK(ctx, tf_malloc((void **)&ctx->d_J, npass * nn * sizeof(double)));
    HIPCHK(ctx, tf_malloc((void **)&ctx->d_K, npass * nn * sizeof(double)));
    HIPCHK(ctx, tf_malloc((void **)&ctx->d_P, npass * nn * sizeof(double)));
    DBG("build_eri done");
    ctx->have_eri = true;
    return TF_OK;
}

int tf_eri_storage(const tf_ctx *ctx, int64_t *bytes, int64_t *n_rows, int32_t *n, int32_t *ld)
{
    if (!ctx || !ctx->have_eri) return TF_EINVAL;
    if (bytes) *bytes = (int64_t)ctx->n_elems * (int64_t)sizeof(double);
    if (n_rows) *n_rows = ctx->n_rows;
    if (n) *n = ctx->N;
    if (ld) *ld = ctx->ld;
    return TF_OK;
}

int tf_set_eri_layout(tf_ctx *ctx, int layout)
{
    if (!ctx || layout < -1 || layout > 2) return TF_EINVAL;
    ctx->layout_req = layout;
    return TF_OK;
}

int tf_eri_layout(const tf_ctx *ctx) { return (ctx && ctx->have_eri) ? ctx->layout : TF_EINVAL; }
int tf_packed_pad(void) { return TF_TRI_PAD; }

int tf_eri_timings(const tf_ctx *ctx, double *s4)
{
    if (!ctx || !s4) return TF_EINVAL;
    std::copy(ctx->eri_seconds, ctx->eri_seconds + 4, s4);
    return TF_OK;
}

int tf_eri_counts(const tf_ctx *ctx, int64_t *c3)
{
    if (!ctx || !c3) return TF_EINVAL;
    for (int i = 0; i < 3; ++i) c3[i] = ctx->eri_counts[i];
    return TF_OK;
}

int tf_eri_flops(const tf_ctx *ctx, double *nominal_flops)
{
    if (!ctx || !nominal_flops) return TF_EINVAL;
    *nominal_flops = ctx->eri_nominal_flops;
    return TF_OK;
}

int tf_segment_pad(void) { return TF_SEG_PAD; }

int tf_copy_eri(tf_ctx *ctx, double *host_out)
{
    if (!ctx) return TF_EINVAL;
    if (!ctx->have_eri || !host_out) TF_FAIL(ctx, TF_EINVAL, "tf_copy_eri: no tensor built / null output");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t total = (size_t)ctx->N * ctx->N * ctx->N * ctx->N;
    double *d_dense = nullptr;
    HIPCHK(ctx, tf_malloc((void **)&d_dense, total * sizeof(double)));
    const unsigned g = (unsigned)std::min<size_t>((total + 255) / 256, 1 << 20);
    if (ctx->layout == 2)
        hipLaunchKernelGGL(expand_dense_tiles_kernel, dim3(g), dim3(256), 0, 0, ctx->d_eri, ctx->tv, ctx->bl, d_dense);
    else if (ctx->layout == 1)
        hipLaunchKernelGGL(expand_dense_packed_kernel, dim3(g), dim3(256), 0, 0, ctx->d_eri, ctx->d_rowmap, ctx->d_rowoff, ctx->d_rowsec, ctx->bl, d_dense);
    else
        hipLaunchKernelGGL(expand_dense_kernel, dim3(g), dim3(256), 0, 0, ctx->d_eri, ctx->d_rowmap, ctx->N, ctx->ld, d_dense);
    hipError_t e = hipMemcpy(host_out, d_dense, total * sizeof(double), hipMemcpyDeviceToHost);
    (void)tf_free(d_dense);
    HIPCHK(ctx, e);
    return TF_OK;
}

int tf_sample_eri(tf_ctx *ctx, int64_t n_idx, const int32_t *idx, double *values)
{
    if (!ctx) return TF_EINVAL;
    if (!ctx->have_eri || !idx || !values || n_idx < 0) TF_FAIL(ctx, TF_EINVAL, "tf_sample_eri: bad arguments");
    for (int64_t q = 0; q < 4 * n_idx; ++q)
        if (idx[q] < 0 || idx[q] >= ctx->N) TF_FAIL(ctx, TF_EINVAL, "tf_sample_eri: index out of range");
    if (n_idx == 0) return TF_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int *d_idx = nullptr; double *d_val = nullptr;
    HIPCHK(ctx, tf_malloc((void **)&d_idx, (size_t)n_idx * 4 * sizeof(int)));
    HIPCHK(ctx, tf_malloc((void **)&d_val, (size_t)n_idx * sizeof(double)));
    HIPCHK(ctx, hipMemcpy(d_idx, idx, (size_t)n_idx * 4 * sizeof(int), hipMemcpyHostToDevice));
    if (ctx->layout == 2)
        hipLaunchKernelGGL(sample_tiles_kernel, dim3((unsigned)((n_idx + 255) / 256)), dim3(256), 0, 0, ctx->d_eri, ctx->tv, ctx->bl, (long long)n_idx, d_idx, d_val);
    else if (ctx->layout == 1)
        hipLaunchKernelGGL(sample_packed_kernel, dim3((unsigned)((n_idx + 255) / 256)), dim3(256), 0, 0, ctx->d_eri, ctx->d_rowmap,
                           ctx->d_rowoff, ctx->d_rowsec, ctx->bl, (long long)n_idx, d_idx, d_val);
    else
        hipLaunchKernelGGL(sample_kernel, dim3((unsigned)((n_idx + 255) / 256)), dim3(256), 0, 0, ctx->d_eri, ctx->d_rowmap, ctx->N,
                           ctx->ld, (long long)n_idx, d_idx, d_val);
    hipError_t e = hipMemcpy(values, d_val, (size_t)n_idx * sizeof(double), hipMemcpyDeviceToHost);
    (void)tf_free(d_idx); (void)tf_free(d_val);
    HIPCHK(ctx, e);
    return TF_OK;
}

// ---- J/K ------------------------------------------------------------------------------------------

// nd = 1 or 2 densities in one pass over the tensor.  dP/dJ/dK: nd dense [N,N] device matrices each.
// One pass of jk_packed_kernel over ND densities (device, symmetric for the exchange part; X = what the D terms contract with) and the
// reductions; dDout[d]: the D matrix of density d.  Tables: set ND - 1.
extern "C++" {
template <int ND>
static int jk_packed_pass(tf_ctx *ctx, hipStream_t st, double *const *dDout, bool cd = false)
{
    const BLayout &L = ctx->bl;
    const int N = ctx->N, NW = L.NW;
    const size_t nn = (size_t)N * N, npr = (size_t)L.NPtot, nrows = (size_t)std::max<long long>(1, ctx->n_rows);
    const tf_ctx::JKTables &T = ctx->jkt[ND - 1];
    const size_t ng = (size_t)std::max(1, T.n_groups);
    // layout of the partial arrays: [column parts of density 0 | .. density 1 | row parts of density 0 | .. density 1]
    JKStrides S{};
    const int MP = ctx->hl.MP;
    S.KS = ctx->hl.KS; S.MP = MP;
    S.planeJd = nrows * NW; S.planeI = ng * N; S.planeJ = nrows * N;
    S.P = nn; S.Pp = npr; S.y = (size_t)T.ypart_len; S.Jd = MP * S.planeJd;
    S.DIc = MP * S.planeI; S.DIr = ng * (size_t)L.RS; S.DJc = MP * S.planeJ; S.DJr = nrows * (size_t)L.RS;
    // region of this pass type inside the partial buffers (tf_build_eri: [one-density pass | two-density pass])
    const size_t per_g = (size_t)N * MP + L.RS;
    // cd: the class-diagonal task list with its own partial-sum buffers (launch_jk_packed decides)
    // (the Jt partials of the two pass types are laid out by their own super-groups: in the general buffer every pass overwrites what it
    // reads, in the class-diagonal buffer the slots of skipped tasks must STAY zero -- each pass type has its own region there)
    double *const pJrow = cd ? ctx->cd_Jrow : ctx->d_Jrow;
    double *const pY = cd ? ctx->cd_ypart + (ND == 2 ? (size_t)std::max<long long>(0, ctx->jkt[0].ypart_len) : 0) : ctx->d_ypart;
    const JKTask *const tasks = cd ? T.d_tasks_cd : T.d_tasks;
    const int n_tasks = cd ? T.n_tasks_cd : T.n_tasks;
    const int *const bucket = cd ? T.bucket_cd : T.bucket;
    double *DI0 = (cd ? ctx->cd_DI : ctx->d_DI) + (ND == 2 ? (size_t)std::max(1, ctx->jkt[0].n_groups) * per_g : 0);
    double *DJ0 = (cd ? ctx->cd_DJ : ctx->d_DJ) + (ND == 2 ? nrows * per_g : 0);
    double *DIc = DI0, *DIr = DI0 + ND * S.DIc, *DJc = DJ0, *DJr = DJ0 + ND * S.DJc;
    if (n_tasks > 0) {
        hipEvent_t ev_after = nullptr;
        if (ctx->prof_jk) {
            if (ctx->prof_used + 2 > ctx->prof_ev.size()) {
                hipEvent_t a, b;
                if (hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) { ctx->prof_ev.push_back(a); ctx->prof_ev.push_back(b); }
            }
            if (ctx->prof_used + 2 <= ctx->prof_ev.size()) {
                (void)hipEventRecord(ctx->prof_ev[ctx->prof_used], st);
                ev_after = ctx->prof_ev[ctx->prof_used + 1];
                ctx->prof_used += 2;
            }
        }
        // one launch per workgroup size; the launches are independent (disjoint tasks): the smaller ones go to side streams so that
        // their ramp-up and tail overlap the big one (fork / join on events; TF_JK_SERIAL=1: all on the caller's stream)
        static const bool serial = getenv("TF_JK_SERIAL") != nullptr;
        int n_launch = 0;
        for (int b = 0; b < 3; ++b) n_launch += (bucket[b + 1] > bucket[b] && (TF_JKP_W >> b) >= 1) ? 1 : 0;
        // Few tasks (small tensors: N2/cc-pVTZ has 1 400): ONE launch of full-size workgroups over all of them -- the idle waves of the
        // narrower tasks cost nothing on a chip the build cannot fill, two launches and the fork / join events of the side streams do
        // (a Fock build at N = 60 is ~90 us of launches, not of work)
        static const int one_launch_below = getenv("TF_JK_ONE_LAUNCH") ? atoi(getenv("TF_JK_ONE_LAUNCH")) : 12000;   // (measured: N = 60 99 -> 69 us per build, 118: 155 -> 108, 160: 202 -> 183, 200: equal, 300: 874 against 907)
        // (one rank only: the rows of a rank of several are short runs of j -- mostly one group per super-group, three of four waves idle
        // in the barriers of a LONG walk; measured at N = 400 on 8 ranks: 0.72 ms per local build in one launch against 0.46 ms in three)
        // (a rank of several only when its super-groups are full -- the whole-shell plan: >= 24 rows per super-group on average; 0.382 -> 0.363 ms
        // per local build on 8 ranks.  Under the segment plan a rank's super-groups hold a few rows each: see above.)
        const bool full_supers = T.n_supers > 0 && ctx->n_rows >= 24LL * T.n_supers;
        const bool one_launch = n_launch > 1 && n_tasks < one_launch_below && (ctx->world == 1 || full_supers);
        const bool fork = !serial && !one_launch && ctx->have_streams && n_launch > 1;
        if (fork) (void)hipEventRecord(ctx->sev[0], st);
        int side = 0;
        bool first = true;
        if (one_launch)
            hipLaunchKernelGGL((jk_packed_kernel<ND>), dim3((unsigned)n_tasks), dim3(64 * TF_JKP_W), 0, st, ctx->d_eri, T.d_groups,
                               T.d_supers, tasks, L, L.kinfo, ctx->d_Psym, ctx->d_Pp, pJrow, pY, DIc, DIr, DJc, DJr, S);
        for (int b = 0; b < 3 && !one_launch; ++b) {
            const int t0 = bucket[b], t1 = bucket[b + 1];
            if (!(t1 > t0 && (TF_JKP_W >> b) >= 1)) continue;
            hipStream_t ls = st;
            if (fork && !first) {
                ++side;
                ls = ctx->streams[side];
                (void)hipStreamWaitEvent(ls, ctx->sev[0], 0);
            }
            first = false;
            hipLaunchKernelGGL((jk_packed_kernel<ND>), dim3((unsigned)(t1 - t0)), dim3(64 * (TF_JKP_W >> b)), 0, ls, ctx->d_eri, T.d_groups,
                               T.d_supers, tasks + t0, L, L.kinfo, ctx->d_Psym, ctx->d_Pp, pJrow, pY, DIc, DIr, DJc, DJr, S);
            if (ls != st) {
                (void)hipEventRecord(ctx->sev[side], ls);
                (void)hipStreamWaitEvent(st, ctx->sev[side], 0);
            }
        }
        if (ev_after) (void)hipEventRecord(ev_after, st);
    }
    JKReduce R{};
    R.ypart = pY; R.sy = S.y; R.supers = T.d_supers; R.MC = ctx->hl.MC; R.MP = MP; R.planeI = S.planeI; R.planeJ = S.planeJ; R.nseg = T.nseg; R.jp = T.jp;
    R.Jt = ctx->d_Jt; R.sJt = (size_t)T.nseg * npr;
    R.DIc = DIc; R.sDIc = S.DIc; R.DIr = DIr; R.sDIr = S.DIr; R.DJc = DJc; R.sDJc = S.DJc; R.DJr = DJr; R.sDJr = S.DJr;
    R.gfirst = T.d_gfirst; R.jptr = ctx->d_jptr; R.jrows = ctx->d_jrows; R.row_ij = ctx->d_row_ij; R.xorder = ctx->d_xorder;
    for (int d = 0; d < ND; ++d) R.D[d] = dDout[d];
    { static const int jkr_dbg = getenv("TF_JKR_DBG") ? atoi(getenv("TF_JKR_DBG")) : 0; R.dbg = jkr_dbg; }
    const unsigned nblk = (unsigned)ND * ((unsigned)N * ((N + 127) / 128) + (unsigned)T.jp.bfirst[4] * T.nseg);
    hipLaunchKernelGGL(jk_reduce_kernel, dim3(nblk), dim3(TF_JKR_THREADS), 0, st, R, L);
    return TF_OK;
}
}  // extern "C++"

// nonsym[d] != 0: density d is not symmetric -- two passes (K = D(P^T) + D(P)^T); nullptr = all symmetric.
static int launch_jk_packed(tf_ctx *ctx, int nd, const double *const *dP, double *const *dJ, double *const *dK, hipStream_t st,
                            const int *nonsym)
{
    const BLayout &L = ctx->bl;
    const int N = ctx->N, NW = L.NW;
    const size_t nn = (size_t)N * N, npr = (size_t)L.NPtot, nrows = (size_t)std::max<long long>(1, ctx->n_rows);
    const dim3 gN((N * N + 255) / 256), b256(256);
    static const bool no_fuse = getenv("TF_JK_NOFUSE") != nullptr;
    // Class-diagonal densities (what an SCF cycle of a diatomic without a transverse field produces: no element between AOs of
    // different x/y parity) need a quarter fewer tasks -- see JKTables::d_tasks_cd.  Only when the caller expects them (the native cycles
    // set jk_try_class_diagonal: the test costs a small kernel and one 16-byte read-back per call) and only when no density of the call
    // has an element between two classes above 1e-14 of its largest element -- the threshold under which the blocked eigensolver
    // (tf_scf.hip.h) already treats such elements as zero.  With exact zeros there (densities built from class-pure orbitals) the skipped
    // products are exact zeros and J, K come out bit for bit the same; DIIS mixtures that carry 1e-16 of a host-made guess differ by that.
    bool cd = false;
    {
        static const bool cd_off = getenv("TF_JK_CLASS_DIAGONAL") && getenv("TF_JK_CLASS_DIAGONAL")[0] == '0';
        bool any_general = false;
        for (int d = 0; d < nd; ++d) any_general = any_general || (nonsym && nonsym[d]);
        // (the test is a kernel and a read-back, ~30 us: a build of a small tensor takes less than that in all -- N2/cc-pVTZ 70 us)
        static const int cd_nmin = getenv("TF_JK_CD_NMIN") ? atoi(getenv("TF_JK_CD_NMIN")) : 160;
        if (ctx->jk_try_class_diagonal && !cd_off && !any_general && N >= cd_nmin && ctx->jkt[0].n_tasks_cd < ctx->jkt[0].n_tasks) {
            if (!ctx->d_cdflag) HIPCHK(ctx, tf_malloc((void **)&ctx->d_cdflag, 2 * sizeof(unsigned long long)));
            cd = true;
            for (int d = 0; d < nd && cd; ++d) {                   // (per density: the threshold is relative to ITS largest element)
                HIPCHK(ctx, hipMemsetAsync(ctx->d_cdflag, 0, 2 * sizeof(unsigned long long), st));
                hipLaunchKernelGGL(class_cross_max_kernel, dim3((unsigned)N), dim3(256), 0, st, dP[d], L, ctx->d_cdflag);
                double h[2] = {0.0, 1.0};
                HIPCHK(ctx, hipMemcpyAsync(h, ctx->d_cdflag, sizeof(h), hipMemcpyDeviceToHost, st));
                HIPCHK(ctx, hipStreamSynchronize(st));
                cd = std::isfinite(h[0]) && h[0] < 1e299 && h[1] <= 1e-14 * h[0];
            }
            if (cd && !ctx->cd_Jrow) {                             // the second set of partial-sum buffers: zeroed once, like the first
                double **bufs[4] = {&ctx->cd_Jrow, &ctx->cd_ypart, &ctx->cd_DI, &ctx->cd_DJ};
                for (int q = 0; q < 4; ++q) {
                    if (tf_malloc((void **)bufs[q], ctx->cd_bytes[q]) != hipSuccess) {     // no room: the full list then
                        (void)hipGetLastError();
                        for (int u = 0; u < q; ++u) { (void)tf_free(*bufs[u]); *bufs[u] = nullptr; }
                        cd = false;
                        break;
                    }
                    HIPCHK(ctx, hipMemsetAsync(*bufs[q], 0, ctx->cd_bytes[q], st));
                }
            }
            if (cd) ++ctx->jk_cd_passes; else ++ctx->jk_cd_declined;
        }
    }
    double *const pJrow = cd ? ctx->cd_Jrow : ctx->d_Jrow;
    if (nd == 2 && !no_fuse && !(nonsym && (nonsym[0] || nonsym[1]))) {
        // two symmetric densities (UHF alpha / beta): one pass over the tensor, groups of 4 rows x 2 densities
        for (int d = 0; d < 2; ++d)
            hipLaunchKernelGGL(pack_density_kernel, gN, b256, 0, st, dP[d], L, 0, ctx->d_Psym + d * nn, ctx->d_Pp + d * npr);
        double *dD[2] = {ctx->d_D, ctx->d_D + nn};
        int rc = jk_packed_pass<2>(ctx, st, dD, cd);
        if (rc) return rc;
        const int nseg = ctx->jkt[1].nseg;
        for (int d = 0; d < 2; ++d)
            hipLaunchKernelGGL(jk_packed_final_kernel, gN, b256, 0, st, dD[d], dD[d], pJrow + d * ctx->hl.MP * nrows * NW, nrows * NW,
                               ctx->hl.KS, ctx->hl.MP, ctx->d_Jt + (size_t)d * nseg * npr, nseg, ctx->d_rowmap, L, dJ[d], dK[d]);
        return TF_OK;
    }
    for (int d = 0; d < nd; ++d)                                 // one density per pass over the packed tensor
      for (int pass = 0; pass < ((nonsym && nonsym[d]) ? 2 : 1); ++pass) {
        const bool general = nonsym && nonsym[d];
        double *dD = (pass == 0) ? ctx->d_D : ctx->d_D + nn;
        hipLaunchKernelGGL(pack_density_kernel, gN, b256, 0, st, dP[d], L, (general && pass == 0) ? 1 : 0, ctx->d_Psym, ctx->d_Pp);
        double *dDp[1] = {dD};
        int rc = jk_packed_pass<1>(ctx, st, dDp, cd);
        if (rc) return rc;
        if (general && pass == 0) continue;
        hipLaunchKernelGGL(jk_packed_final_kernel, gN, b256, 0, st, ctx->d_D, dD, pJrow, nrows * NW, ctx->hl.KS, ctx->hl.MP, ctx->d_Jt,
                           ctx->jkt[0].nseg, ctx->d_rowmap, L, dJ[d], dK[d]);
      }
    return TF_OK;
}

// Fock build from the tiles layout (tf_jktile.hip.h): pack -> jk_tile_kernel beside jk_edge_kernel -> jk_tile_reduce_kernel ->
// jk_tile_final_kernel.  One density per pass (a non-symmetric one: two passes, K = D(P^T) + D(P)^T), or -- symmetric densities only --
// up to eight per pass: densities = columns of the B operands of the matrix-core products (ND = 4 or 8; fewer: zero columns).
extern "C++" {
// the partial-sum buffers of a list for nd densities per pass
static int tile_list_buffers(tf_ctx *ctx, tf_ctx::TileList &D, int nd)
{
    if (D.nd_cap >= nd) return TF_OK;
    auto alloc_t = [&](double **p, size_t doubles) -> int {
        HIPCHK(ctx, tf_malloc((void **)p, std::max<size_t>(1, doubles) * sizeof(double)));
        ctx->tile_allocs.push_back(*p);
        return TF_OK;
    };
    int rc;
    if ((rc = alloc_t(&D.DJ, (size_t)D.dj_len * nd)) || (rc = alloc_t(&D.Jt, (size_t)D.jt_len * nd)) || (rc = alloc_t(&D.Jd, (size_t)D.jd_len * nd)) ||
        (rc = alloc_t(&D.DIk, (size_t)D.n_di * 64 * nd)) || (rc = alloc_t(&D.DIl, (size_t)D.n_di * 16 * nd)))
        return rc;
    D.nd_cap = nd;
    return TF_OK;
}
// device copy of a task list
static int tile_list_upload(tf_ctx *ctx, const tft::TaskList &TL, bool own_shapes, tf_ctx::TileList &D)
{
    auto up_t = [&](const auto &h, auto **d) -> int {
        int rc3 = upload(ctx, h, d, false);
        if (!rc3) ctx->tile_allocs.push_back(*d);
        return rc3;
    };
    int rc;
    if (own_shapes && ((rc = up_t(TL.pairs, &D.d_pairs)) || (rc = up_t(TL.runs, &D.d_runs)))) return rc;   // (the shapes of the DJ vectors differ with the strip height)
    if ((rc = up_t(TL.tasks, &D.d_tasks)) || (rc = up_t(TL.itask_ptr, &D.d_itask_ptr)) || (rc = up_t(TL.itasks, &D.d_itasks))) return rc;
    D.n_tasks = (int)TL.tasks.size(); D.ksub = TL.ksub; D.n_di = TL.n_di; D.dj_len = TL.dj_len; D.jd_len = TL.jd_len; D.jt_len = TL.jt_len;
    for (int b = 0; b <= TT_W; ++b) D.bucket[b] = TL.bucket[b];
    return TF_OK;
}

// one pass over the tensor for the ND densities in ctx->t_X / t_Pm (packed by the caller); `out`: [6][ND][N][N]
template <int ND>
static int jk_tiles_pass(tf_ctx *ctx, tf_ctx::TileList &D, hipStream_t st, double *out)
{
    const int N = ctx->N;
    const size_t nn = (size_t)N * N;
    const tft::Tables &TT = ctx->tiles;
    int rc = tile_list_buffers(ctx, D, ND);
    if (rc) return rc;
    int jt_rows = 0;
    for (int p = 0; p < TT.npair; ++p) jt_rows += ctx->hl.csize[TT.pa[p]];
    const size_t edge_lds = (size_t)(2 + 3 * TT_EDGE_WAVES) * N * sizeof(double), red_lds = std::max((size_t)2 * TT_RED_WAVES * N, (size_t)TT_RED_WAVES * 64 * std::max(TT_RED_CT, ND)) * sizeof(double);
    if (std::max(edge_lds, red_lds) > ctx->tile_lds_set) {                // (beyond the default 64 KB per workgroup: N > 580)
        if (std::max(edge_lds, red_lds) > (size_t)160 * 1024 - 1024) TF_FAIL(ctx, TF_EINVAL, "N = %d exceeds the LDS rows of the tiles layout's reductions", N);
        const int want = (int)std::max(edge_lds, red_lds);
        HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&jk_edge_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, want));
        HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&jk_edge_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, want));
        HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&jk_edge_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, want));
        HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&jk_tile_reduce_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, want));
        HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&jk_tile_reduce_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, want));
        HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&jk_tile_reduce_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, want));
        ctx->tile_lds_set = (size_t)want;
    }
    TJArgs A{};
    A.DJ = D.DJ; A.Jt = D.Jt; A.Jd = D.Jd; A.DIk = D.DIk; A.DIl = D.DIl; A.sJt = (size_t)D.jt_len; A.sDIk = (size_t)D.n_di * 64; A.sDIl = (size_t)D.n_di * 16;
    A.N = N; A.pm_len = TT.pm_len;
    TEArgs E{};
    E.edge_base = TT.edge_base; E.N = N; E.tab = ctx->d_tvtab; E.EJ = out + 5 * ND * nn; E.ED = out + 2 * ND * nn; E.EDT = out + 3 * ND * nn; E.sE = nn;
    const bool fork = ctx->have_streams && getenv("TF_JK_SERIAL") == nullptr;
    if (D.n_tasks > 0) {
        hipEvent_t ev_after = nullptr;
        if (ctx->prof_jk) {
            if (ctx->prof_used + 2 > ctx->prof_ev.size()) {
                hipEvent_t a, b;
                if (hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) { ctx->prof_ev.push_back(a); ctx->prof_ev.push_back(b); }
            }
            if (ctx->prof_used + 2 <= ctx->prof_ev.size()) {
                (void)hipEventRecord(ctx->prof_ev[ctx->prof_used], st);
                ev_after = ctx->prof_ev[ctx->prof_used + 1];
                ctx->prof_used += 2;
            }
        }
        if (fork) (void)hipEventRecord(ctx->sev[0], st);
        // ONE launch of full-size workgroups over all tasks (measured at N = 400: 1.57 ms against 1.82 for one launch per workgroup size
        // on side streams -- loads only); the idle waves of the narrower tasks wait in the task's barriers
        static const int pf_env = getenv("TF_TILE_PF") ? atoi(getenv("TF_TILE_PF")) : 2;
        const dim3 grid((unsigned)D.n_tasks), block(64 * TT_W);
#define TF_TJ_LAUNCH(NDV, MBV, PFV) hipLaunchKernelGGL((jk_tile_kernel<NDV, MBV, PFV>), grid, block, 0, st, ctx->d_eri, D.d_tasks, ctx->t_X, ctx->t_Pm, A)
        if constexpr (ND == 1) {
            if (D.ksub == 16) { if (pf_env == 1) TF_TJ_LAUNCH(1, 1, 1); else TF_TJ_LAUNCH(1, 1, 2); }
            else if (D.ksub == 32) { if (pf_env == 1) TF_TJ_LAUNCH(1, 2, 1); else TF_TJ_LAUNCH(1, 2, 2); }
            else TF_TJ_LAUNCH(1, 4, 1);
        } else {
            if (D.ksub != 16) TF_FAIL(ctx, TF_EINVAL, "tiles layout: the wide pass needs the list of 16-row strips");
            if (pf_env == 1) TF_TJ_LAUNCH(ND, 1, 1); else TF_TJ_LAUNCH(ND, 1, 2);
        }
#undef TF_TJ_LAUNCH
        if (ev_after) (void)hipEventRecord(ev_after, st);
    }
    {
        // the edge elements beside the tile launch
        hipStream_t ls = st;
        if (fork && D.n_tasks > 0) { ls = ctx->streams[1]; (void)hipStreamWaitEvent(ls, ctx->sev[0], 0); }
        hipLaunchKernelGGL((jk_edge_kernel<ND>), dim3((unsigned)N, (unsigned)ND), dim3(64 * TT_EDGE_WAVES), edge_lds, ls, ctx->d_eri, ctx->t_X, D.d_runs, E);
        if (ls != st) { (void)hipEventRecord(ctx->sev[1], ls); (void)hipStreamWaitEvent(st, ctx->sev[1], 0); }
    }
    TRArgs R{};
    R.itask_ptr = D.d_itask_ptr; R.itasks = D.d_itasks; R.jlist_ptr = ctx->d_jlist_ptr; R.clsI = ctx->bl.clsI;
    R.DJ = D.DJ; R.Jt = D.Jt; R.Jd = D.Jd; R.DIk = D.DIk; R.DIl = D.DIl; R.T = ctx->d_eri; R.X = ctx->t_X;
    R.sJt = A.sJt; R.sDIk = A.sDIk; R.sDIl = A.sDIl; R.sO = nn;
    R.Dj = out; R.Di = out + (size_t)ND * nn; R.JD = out + (size_t)4 * ND * nn; R.JtTot = ctx->t_JtTot;
    R.edge_base = TT.edge_base; R.N = N; R.ksub = D.ksub; R.npair = TT.npair; R.nd = ND; R.pm_len = TT.pm_len; R.tab = ctx->d_tvtab;
    R.jt_rows = jt_rows;
    const unsigned nblk = ND == 1 ? (unsigned)(5 * N + jt_rows) : (unsigned)(4 * N) + (unsigned)ND * (unsigned)(N + jt_rows);
    hipLaunchKernelGGL(jk_tile_reduce_kernel<ND>, dim3(nblk), dim3(TT_RED_THREADS), red_lds, st, D.d_tasks, D.d_pairs, D.d_runs, ctx->d_jlist, R);
    return TF_OK;
}
}  // extern "C++"

static int launch_jk_tiles(tf_ctx *ctx, int nd, const double *const *dP, double *const *dJ, double *const *dK, hipStream_t st, const int *nonsym)
{
    const int N = ctx->N;
    const size_t nn = (size_t)N * N, pml = (size_t)std::max(1, ctx->tiles.pm_len);
    const dim3 gN((unsigned)((nn + 255) / 256)), b256(256);
    auto final_launch = [&](const double *o0, const double *o1, int ndp, int d, const double *jtt, double *J, double *K) {
        TFArgs F{};                                                          // out sets: [6][ndp][N][N]; D of the first pass, D2 of the last
        F.Dj = o0 + (size_t)(0 * ndp + d) * nn; F.Di = o0 + (size_t)(1 * ndp + d) * nn; F.ED = o0 + (size_t)(2 * ndp + d) * nn; F.EDT = o0 + (size_t)(3 * ndp + d) * nn;
        F.Dj2 = o1 + (size_t)(0 * ndp + d) * nn; F.Di2 = o1 + (size_t)(1 * ndp + d) * nn; F.ED2 = o1 + (size_t)(2 * ndp + d) * nn; F.EDT2 = o1 + (size_t)(3 * ndp + d) * nn;
        F.JD = o1 + (size_t)(4 * ndp + d) * nn; F.EJ = o1 + (size_t)(5 * ndp + d) * nn; F.JtTot = jtt; F.tab = ctx->d_tvtab;
        hipLaunchKernelGGL(jk_tile_final_kernel, gN, b256, 0, st, F, ctx->bl, J, K);
    };
    bool any_general = false;
    for (int d = 0; d < nd; ++d) any_general = any_general || (nonsym && nonsym[d]);
    static const int wide_min = getenv("TF_TILE_WIDE_MIN") ? atoi(getenv("TF_TILE_WIDE_MIN")) : 2;     // fewest densities that go through the wide pass
    if (nd >= wide_min && nd >= 2 && !any_general) {
        // the list of 16-row strips (one row block per wave: the registers hold ND accumulator sets), built at first use
        if (!ctx->tlw.d_tasks) {
            static const int part_steps = std::min(TT_STEPS_MAX, getenv("TF_TILE_PART_STEPS") ? std::max(1, atoi(getenv("TF_TILE_PART_STEPS"))) : TT_STEPS_MAX);
            const tft::TaskList *TL = &ctx->tsubw;
            if (ctx->ksub1 == 16) TL = &ctx->tsub1;
            else {
                const std::string e = tft::build_list(ctx->tclass, ctx->tiles, ctx->trows, 16, part_steps, ctx->tsubw);
                if (!e.empty()) TF_FAIL(ctx, TF_EINVAL, "%s", e.c_str());
            }
            int rc = tile_list_upload(ctx, *TL, true, ctx->tlw);
            if (rc) return rc;
        }
        for (int d0 = 0; d0 < nd;) {
            const int left = nd - d0, ndp = left > 4 ? 8 : 4, n = std::min(left, ndp);
            for (int d = 0; d < n; ++d)
                hipLaunchKernelGGL(pack_density_tiles_kernel, gN, b256, 0, st, dP[d0 + d], ctx->bl, ctx->tv, 0, ctx->t_X + (size_t)d * nn, ctx->t_Pm + (size_t)d * pml);
            if (n < ndp) {                                                   // empty columns
                HIPCHK(ctx, hipMemsetAsync(ctx->t_X + (size_t)n * nn, 0, (size_t)(ndp - n) * nn * sizeof(double), st));
                HIPCHK(ctx, hipMemsetAsync(ctx->t_Pm + (size_t)n * pml, 0, (size_t)(ndp - n) * pml * sizeof(double), st));
            }
            int rc = ndp == 8 ? jk_tiles_pass<8>(ctx, ctx->tlw, st, ctx->t_out) : jk_tiles_pass<4>(ctx, ctx->tlw, st, ctx->t_out);
            if (rc) return rc;
            for (int d = 0; d < n; ++d) final_launch(ctx->t_out, ctx->t_out, ndp, d, ctx->t_JtTot + (size_t)d * pml, dJ[d0 + d], dK[d0 + d]);
            d0 += n;
        }
        return TF_OK;
    }
    for (int d = 0; d < nd; ++d) {
        const bool general = nonsym && nonsym[d];
        for (int pass = 0; pass < (general ? 2 : 1); ++pass) {
            double *out = ctx->t_out + (size_t)pass * 6 * nn;
            hipLaunchKernelGGL(pack_density_tiles_kernel, gN, b256, 0, st, dP[d], ctx->bl, ctx->tv, (general && pass == 0) ? 1 : 0, ctx->t_X, ctx->t_Pm);
            int rc = jk_tiles_pass<1>(ctx, ctx->tl1, st, out);
            if (rc) return rc;
            if (general && pass == 0) continue;
            final_launch(ctx->t_out, out, 1, 0, ctx->t_JtTot, dJ[d], dK[d]);
        }
    }
    return TF_OK;
}

static int launch_jk(tf_ctx *ctx, int nd, const double *const *dP, double *const *dJ, double *const *dK, hipStream_t st,
                     const int *nonsym = nullptr)
{
    if (ctx->layout == 2) return launch_jk_tiles(ctx, nd, dP, dJ, dK, st, nonsym);
    if (ctx->layout == 1) return launch_jk_packed(ctx, nd, dP, dJ, dK, st, nonsym);
    const int N = ctx->N, ld = ctx->ld;
    const double *Ppad[2] = {dP[0], nd > 1 ? dP[1] : dP[0]};
    if (ld != N) {
        for (int d = 0; d < nd; ++d) {
            double *dst = ctx->d_Ppad + (size_t)d * N * ld;
            hipLaunchKernelGGL(pad_matrix_kernel, dim3((N * ld + 255) / 256), dim3(256), 0, st, dP[d], dst, N, ld);
            Ppad[d] = dst;
        }
    }
    if (ctx->n_rows > 0) {
        const int npair = ld / 2;
        // rows per workgroup: share each P tile among JB rows, but keep >= ~2 workgroups per CU in flight
        int JB = (ctx->n_rows >= 4 * 2048) ? 4 : (ctx->n_rows >= 2 * 2048 ? 2 : 1);
        if (nd == 2 && JB == 4) JB = 2;                         // register budget
        int nlc = (npair <= TF_JK_THREADS) ? 1 : (npair <= 2 * TF_JK_THREADS ? 2 : 4);
        // test hooks (tests/test_gpu_sharded.py): exercise the variants that only N > 512 would select
        if (const char *e = getenv("TF_JK_FORCE_NLC")) nlc = std::max(nlc, atoi(e));
        if (const char *e = getenv("TF_JK_FORCE_JB")) { const int f = atoi(e); if (f == 1 || f == 2 || (f == 4 && nd == 1)) JB = f; }
        // instantiated combinations: (NLC 4, one density) up to JB 2; (NLC 4, two densities) JB 1 -- the grid below must match
        if (nlc == 4 && nd == 1 && JB == 4) JB = 2;
        if (nlc == 4 && nd == 2) JB = 1;
        const size_t smem = (size_t)(2 * nd * JB * N + 4 * TF_JK_THREADS) * sizeof(double);
        const dim3 grid((unsigned)((ctx->n_rows + JB - 1) / JB)), block(TF_JK_THREADS);
        hipEvent_t ev_after = nullptr;
        if (ctx->prof_jk) {
            if (ctx->prof_used + 2 > ctx->prof_ev.size()) {
                hipEvent_t a, b;
                if (hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) { ctx->prof_ev.push_back(a); ctx->prof_ev.push_back(b); }
            }
            if (ctx->prof_used + 2 <= ctx->prof_ev.size()) {
                (void)hipEventRecord(ctx->prof_ev[ctx->prof_used], st);
                ev_after = ctx->prof_ev[ctx->prof_used + 1];
                ctx->prof_used += 2;
            }
        }
#define TF_JK_LAUNCH(NLC, JBV, NDV)                                                                                         \
        hipLaunchKernelGGL((jk_rows_kernel<NLC, JBV, NDV>), grid, block, smem, st, ctx->d_eri, ctx->d_row_ij, ctx->n_rows, N, ld, \
                           Ppad[0], Ppad[1], ctx->d_Jrow, ctx->d_Kp)
        if (npair > 4 * TF_JK_THREADS) TF_FAIL(ctx, TF_EINVAL, "N = %d exceeds the J/K kernel's row length limit (2048)", N);
        if (nd == 1) {
            if (nlc == 1) {
                if (JB == 4) TF_JK_LAUNCH(1, 4, 1); else if (JB == 2) TF_JK_LAUNCH(1, 2, 1); else TF_JK_LAUNCH(1, 1, 1);
            } else if (nlc == 2) {
                if (JB == 4) TF_JK_LAUNCH(2, 4, 1); else if (JB == 2) TF_JK_LAUNCH(2, 2, 1); else TF_JK_LAUNCH(2, 1, 1);
            } else {
                if (JB >= 2) TF_JK_LAUNCH(4, 2, 1); else TF_JK_LAUNCH(4, 1, 1);
            }
        } else {
            if (nlc == 1) {
                if (JB == 2) TF_JK_LAUNCH(1, 2, 2); else TF_JK_LAUNCH(1, 1, 2);
            } else if (nlc == 2) {
                if (JB == 2) TF_JK_LAUNCH(2, 2, 2); else TF_JK_LAUNCH(2, 1, 2);
            } else
                TF_JK_LAUNCH(4, 1, 2);
        }
#undef TF_JK_LAUNCH
        if (ev_after) (void)hipEventRecord(ev_after, st);
    }
    for (int d = 0; d < nd; ++d)
        hipLaunchKernelGGL(jk_reduce_kernel, dim3(N, (N + 63) / 64), dim3(256), 0, st, ctx->d_Jrow + (size_t)d * ctx->n_rows,
                           ctx->d_Kp + (size_t)d * ctx->n_rows * 2 * ld, ctx->d_rowmap, N, ld, dJ[d], dK[d]);
    return TF_OK;
}

// The single exchange step of a sharded Fock build: the partial J and K of nd densities are stacked in one
// staging buffer [2][nd][N][N] and summed over the ranks by the caller's all-reduce (torch.distributed over RCCL in tuna_amd).
static int allreduce_jk(tf_ctx *ctx, int nd, double *const *dJ, double *const *dK, hipStream_t st)
{
    if (ctx->world == 1) return TF_OK;
    if (ctx->comm) {
        // in the library: ncclAllReduce of J and of K where they lie, one group, on the build's stream (no staging copy, no status word: a
        // failed rank fails the collective for every rank)
        const size_t nn1 = (size_t)ctx->N * ctx->N;
        int grc = tfrccl::GroupStart();
        for (int d = 0; d < nd && !grc; ++d) {
            grc = tfrccl::AllReduce(dJ[d], dJ[d], nn1, tfrccl::kFloat64, tfrccl::kSum, ctx->comm, st);
            if (!grc) grc = tfrccl::AllReduce(dK[d], dK[d], nn1, tfrccl::kFloat64, tfrccl::kSum, ctx->comm, st);
        }
        const int erc = tfrccl::GroupEnd();
        if (grc || erc) TF_FAIL(ctx, TF_ENODEVICE, "ncclAllReduce of [J;K] failed: %s", tfrccl::errstr(grc ? grc : erc).c_str());
        return TF_OK;
    }
    if (!ctx->allreduce)
        TF_FAIL(ctx, TF_EINVAL, "the tensor is sharded over %d ranks: attach an RCCL communicator (tf_comm_init) or register the all-reduce of the partial [J;K] with tf_set_allreduce", ctx->world);
    const size_t nn = (size_t)ctx->N * ctx->N, need = 2 * (size_t)nd * nn;
    // one slot beyond the payload carries a status word through the same collective: a rank whose exchange step failed locally (the
    // hook sets it) makes the sum non-zero on EVERY rank, so that all of them return an error instead of some waiting in a collective
    if (ctx->jkstage_doubles < need + 1) {
        if (ctx->d_jkstage) (void)tf_free(ctx->d_jkstage);
        ctx->d_jkstage = nullptr; ctx->jkstage_doubles = 0;
        HIPCHK(ctx, tf_malloc((void **)&ctx->d_jkstage, (need + 1) * sizeof(double)));
        ctx->jkstage_doubles = need + 1;
    }
    for (int d = 0; d < nd; ++d) {
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_jkstage + (size_t)d * nn, dJ[d], nn * sizeof(double), hipMemcpyDeviceToDevice, st));
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_jkstage + (size_t)(nd + d) * nn, dK[d], nn * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    HIPCHK(ctx, hipMemsetAsync(ctx->d_jkstage + need, 0, sizeof(double), st));
    const int rc = ctx->allreduce(ctx->allreduce_user, ctx->d_jkstage, (int64_t)(need + 1), (void *)st);
    if (rc) TF_FAIL(ctx, TF_ENODEVICE, "the registered all-reduce failed (code %d)", rc);
    double status = 0.0;
    HIPCHK(ctx, hipMemcpyAsync(&status, ctx->d_jkstage + need, sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    if (status != 0.0) TF_FAIL(ctx, TF_ENODEVICE, "the exchange step of the sharded Fock build failed on %g rank(s): every rank stops", status);
    for (int d = 0; d < nd; ++d) {
        HIPCHK(ctx, hipMemcpyAsync(dJ[d], ctx->d_jkstage + (size_t)d * nn, nn * sizeof(double), hipMemcpyDeviceToDevice, st));
        HIPCHK(ctx, hipMemcpyAsync(dK[d], ctx->d_jkstage + (size_t)(nd + d) * nn, nn * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    return TF_OK;
}

// Collective control flow of the native cycles on a sharded tensor (tfscf::Workspace::agree): the values are summed over the ranks
// through the registered all-reduce; a rank whose values differ from the mean makes EVERY rank fail instead of leaving some of them
// waiting in the next collective.
static int agree_over_ranks(tf_ctx *ctx, const double *vals, int n, std::string &msg)
{
    if (ctx->world == 1 || (!ctx->allreduce && !ctx->comm)) return TF_OK;
    if (n > 7) n = 7;
    double h[16] = {0};
    h[0] = 1.0;
    for (int k = 0; k < n; ++k) { h[1 + k] = vals[k]; h[8 + k] = vals[k] * vals[k]; }
    if (!ctx->d_agree && tf_malloc((void **)&ctx->d_agree, 16 * sizeof(double)) != hipSuccess) { msg = "hipMalloc failed (agree buffer)"; return TF_ENOMEM; }
    if (hipMemcpy(ctx->d_agree, h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess) { msg = "hipMemcpy failed (agree buffer)"; return TF_ENODEVICE; }
    int rc;
    if (ctx->comm) { rc = comm_allreduce(ctx, ctx->d_agree, 16, nullptr); if (rc) { msg = ctx->err; return rc; } }
    else rc = ctx->allreduce(ctx->allreduce_user, ctx->d_agree, 16, nullptr);
    if (rc) { msg = "the registered all-reduce failed (code " + std::to_string(rc) + ")"; return TF_ENODEVICE; }
    double s[16];
    if (hipMemcpy(s, ctx->d_agree, sizeof(s), hipMemcpyDeviceToHost) != hipSuccess) { msg = "hipMemcpy failed (agree buffer)"; return TF_ENODEVICE; }
    // slot 15 is the status word of the hook's convention (distributed.py: a rank whose staging failed adds 1 to the last element)
    if (s[15] != 0.0) { msg = "the exchange of the agreement vector failed on at least one rank"; return TF_ENODEVICE; }
    const double w = (double)ctx->world;
    bool same = s[0] == w;
    // all equal <=> sum of squares == (sum)^2 / world (small integers and flags: exact in double precision)
    for (int k = 0; k < n && same; ++k) same = (s[1 + k] == w * vals[k]) && (s[8 + k] == w * vals[k] * vals[k]);
    if (!same) {
        msg = "the ranks of the sharded SCF cycle disagree on a control decision (convergence / DIIS history / step): every rank stops";
        return TF_ELINALG;
    }
    return TF_OK;
}

// Debug / test aid: copies of the partial-sum buffers of the last packed J/K pass and of its group table (one density):
// which = 0 DIc [groups][N], 1 DIr [groups][RS], 2 DJc [rows][N], 3 DJr [rows][RS]; returns the number of doubles copied (<= n) or < 0.
long long tf_debug_partials(tf_ctx *ctx, int which, double *host, long long n)
{
    if (!ctx || !ctx->have_eri || ctx->layout != 1 || !host) return TF_EINVAL;
    const tf_ctx::JKTables &T = ctx->jkt[0];
    const size_t ng = (size_t)std::max(1, T.n_groups), nrows = (size_t)std::max<long long>(1, ctx->n_rows), N = (size_t)ctx->N, RS = (size_t)ctx->bl.RS;
    const double *src = nullptr; size_t cnt = 0;
    switch (which) {
    case 0: src = ctx->d_DI; cnt = ng * N; break;
    case 1: src = ctx->d_DI + ng * N; cnt = ng * RS; break;
    case 2: src = ctx->d_DJ; cnt = nrows * N; break;
    case 3: src = ctx->d_DJ + nrows * N; cnt = nrows * RS; break;
    case 4: src = ctx->d_D; cnt = N * N; break;
    default: return TF_EINVAL;
    }
    cnt = std::min<size_t>(cnt, (size_t)n);
    if (hipMemcpy(host, src, cnt * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return TF_ENODEVICE;
    return (long long)cnt;
}
// group table of the one-density pass: per group {i, j0, nr, r0, c} (internal indices); returns the number of groups
int tf_debug_groups(tf_ctx *ctx, int32_t *out5, int max_groups)
{
    if (!ctx || !ctx->have_eri || ctx->layout != 1) return TF_EINVAL;
    const tf_ctx::JKTables &T = ctx->jkt[0];
    std::vector<JKGroup> g((size_t)T.n_groups);
    if (T.n_groups && hipMemcpy(g.data(), T.d_groups, g.size() * sizeof(JKGroup), hipMemcpyDeviceToHost) != hipSuccess) return TF_ENODEVICE;
    for (int k = 0; k < T.n_groups && k < max_groups; ++k) { out5[5 * k] = g[k].i; out5[5 * k + 1] = g[k].j0; out5[5 * k + 2] = g[k].nr; out5[5 * k + 3] = g[k].r0; out5[5 * k + 4] = g[k].c; }
    return T.n_groups;
}

int tf_set_allreduce(tf_ctx *ctx, tf_allreduce_fn fn, void *user)
{
    if (!ctx) return TF_EINVAL;
    ctx->allreduce = fn; ctx->allreduce_user = user;
    return TF_OK;
}

int tf_comm_unique_id(void *id_out)
{
    if (!id_out) return TF_EINVAL;
    std::string err;
    if (!tfrccl::load(err)) { g_create_error = err; return TF_ENODEVICE; }
    tfrccl::UniqueId id;
    const int rc = tfrccl::GetUniqueId(&id);
    if (rc) { g_create_error = "ncclGetUniqueId failed: " + tfrccl::errstr(rc); return TF_ENODEVICE; }
    std::memcpy(id_out, &id, sizeof(id));
    return TF_OK;
}

int tf_comm_init(tf_ctx *ctx, const void *id, int comm_rank, int comm_size)
{
    if (!ctx || !id || comm_size < 1 || comm_rank < 0 || comm_rank >= comm_size) return TF_EINVAL;
    std::string err;
    if (!tfrccl::load(err)) TF_FAIL(ctx, TF_ENODEVICE, "%s", err.c_str());
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (ctx->comm) { (void)tfrccl::CommDestroy(ctx->comm); ctx->comm = nullptr; }
    tfrccl::UniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    void *comm = nullptr;
    const int rc = tfrccl::CommInitRank(&comm, comm_size, uid, comm_rank);
    if (rc) TF_FAIL(ctx, TF_ENODEVICE, "ncclCommInitRank failed: %s", tfrccl::errstr(rc).c_str());
    ctx->comm = comm;
    return TF_OK;
}

int tf_comm_destroy(tf_ctx *ctx)
{
    if (!ctx) return TF_EINVAL;
    if (ctx->comm) {
        (void)hipSetDevice(ctx->device);
        (void)hipDeviceSynchronize();
        (void)tfrccl::CommDestroy(ctx->comm);
        ctx->comm = nullptr;
    }
    return TF_OK;
}

int tf_comm_attached(const tf_ctx *ctx) { return (ctx && ctx->comm) ? 1 : 0; }

int tf_fock_jk_device(tf_ctx *ctx, int n_dens, const double *dP, double *dJ, double *dK, void *stream)
{
    if (!ctx) return TF_EINVAL;
    if (!ctx->have_eri) TF_FAIL(ctx, TF_EINVAL, "tf_fock_jk: call tf_build_eri first");
    if (n_dens < 1 || !dP || !dJ || !dK) TF_FAIL(ctx, TF_EINVAL, "tf_fock_jk: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t nn = (size_t)ctx->N * ctx->N;
    const int chunk = ctx->layout == 2 ? 8 : 2;                  // densities per pass over the tensor: two (packed / rows), up to eight (tiles)
    for (int d = 0; d < n_dens; d += chunk) {
        const int nd = std::min(chunk, n_dens - d);
        const double *p[8]; double *j[8], *k[8];
        for (int q = 0; q < 8; ++q) { const int dq = d + std::min(q, nd - 1); p[q] = dP + dq * nn; j[q] = dJ + dq * nn; k[q] = dK + dq * nn; }
        int rc = launch_jk(ctx, nd, p, j, k, (hipStream_t)stream);
        if (rc) return rc;
        if (ctx->comm && ctx->world > 1 && (rc = allreduce_jk(ctx, nd, j, k, (hipStream_t)stream))) return rc;   // in the library, on the same stream
    }
    return TF_OK;
}

int tf_fock_jk(tf_ctx *ctx, int n_dens, const double *P, double *J, double *K)
{
    if (!ctx) return TF_EINVAL;
    if (!ctx->have_eri) TF_FAIL(ctx, TF_EINVAL, "tf_fock_jk: call tf_build_eri first");
    if (n_dens < 1 || !P || !J || !K) TF_FAIL(ctx, TF_EINVAL, "tf_fock_jk: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t nn = (size_t)ctx->N * ctx->N;
    const int chunk = ctx->layout == 2 ? 8 : 2;
    for (int d = 0; d < n_dens; d += chunk) {
        const int nd = std::min(chunk, n_dens - d);
        HIPCHK(ctx, hipMemcpy(ctx->d_P, P + d * nn, nd * nn * sizeof(double), hipMemcpyHostToDevice));
        const double *p[8]; double *j[8], *k[8];
        for (int q = 0; q < 8; ++q) { const int dq = std::min(q, nd - 1); p[q] = ctx->d_P + dq * nn; j[q] = ctx->d_J + dq * nn; k[q] = ctx->d_K + dq * nn; }
        int nonsym[8] = {0, 0, 0, 0, 0, 0, 0, 0};                // the packed / tiles layouts need a second pass for a non-symmetric density
        const int N = ctx->N;
        for (int q = 0; q < nd; ++q) {
            const double *Pq = P + (d + q) * nn;
            for (int a = 0; a < N && !nonsym[q]; ++a)
                for (int b = 0; b < a; ++b)
                    if (Pq[(size_t)a * N + b] != Pq[(size_t)b * N + a]) { nonsym[q] = 1; break; }
        }
        int rc = launch_jk(ctx, nd, p, j, k, 0, nonsym);
        if (rc) return rc;
        if (ctx->comm && ctx->world > 1 && (rc = allreduce_jk(ctx, nd, j, k, 0))) return rc;     // a communicator is attached: the sums over the ranks
        HIPCHK(ctx, hipMemcpy(J + d * nn, ctx->d_J, nd * nn * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(K + d * nn, ctx->d_K, nd * nn * sizeof(double), hipMemcpyDeviceToHost));
    }
    HIPCHK(ctx, hipGetLastError());
    return TF_OK;
}

int tf_jk_profile(tf_ctx *ctx, int enable)
{
    if (!ctx) return TF_EINVAL;
    ctx->prof_jk = enable != 0;
    ctx->prof_used = 0;
    return TF_OK;
}

int tf_jk_profile_read(tf_ctx *ctx, double *seconds_total, int64_t *launches)
{
    if (!ctx || !seconds_total || !launches) return TF_EINVAL;
    double tot = 0.0;
    for (size_t k = 0; k + 1 < ctx->prof_used; k += 2) {
        HIPCHK(ctx, hipEventSynchronize(ctx->prof_ev[k + 1]));
        float ms = 0.f;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->prof_ev[k], ctx->prof_ev[k + 1]));
        tot += ms * 1e-3;
    }
    *seconds_total = tot;
    *launches = (int64_t)(ctx->prof_used / 2);
    ctx->prof_used = 0;
    return TF_OK;
}

// ---- one-electron integrals -------------------------------------------------------------------------

int tf_one_electron(tf_ctx *ctx, int n_atoms, const double *atom_xyz, const double *atom_charge, const double *dipole_origin,
                    int spherical, double *S, double *T, double *V, double *D, double *Q)
{
    if (!ctx) return TF_EINVAL;
    if (!ctx->have_basis) TF_FAIL(ctx, TF_EINVAL, "tf_one_electron: call tf_set_basis first");
    if (n_atoms < 1 || n_atoms > 2 || !atom_xyz || !atom_charge || !dipole_origin || !S || !T || !V)
        TF_FAIL(ctx, TF_EINVAL, "tf_one_electron: bad arguments");
    for (int a = 0; a < n_atoms; ++a)
        if (atom_xyz[3 * a] != 0.0 || atom_xyz[3 * a + 1] != 0.0)
            TF_FAIL(ctx, TF_EGEOM, "Molecule is incorrectly aligned! Unable to calculate molecular integrals.");
    if (spherical && !ctx->bs.all_full) TF_FAIL(ctx, TF_EINVAL, "spherical output needs complete shells");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::string msg = tfone::one_electron(ctx->bs, n_atoms, atom_xyz, atom_charge, dipole_origin, spherical, S, T, V, D, Q, ctx->db.boys, &ctx->arena1e);
    if (!msg.empty()) TF_FAIL(ctx, TF_ENODEVICE, "%s", msg.c_str());
    return TF_OK;
}

int tf_cross_overlap(tf_ctx *ctx, int n2, const double *origin2, const int32_t *lmn2, const int32_t *prim_off2,
                     const double *exps2, const double *coefs_raw2, double *S_cross)
{
    if (!ctx) return TF_EINVAL;
    if (!ctx->have_basis) TF_FAIL(ctx, TF_EINVAL, "tf_cross_overlap: call tf_set_basis first");
    if (n2 < 1 || !origin2 || !lmn2 || !prim_off2 || !exps2 || !coefs_raw2 || !S_cross)
        TF_FAIL(ctx, TF_EINVAL, "tf_cross_overlap: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    tf::Basis other;
    std::string msg = tf::build_basis(other, n2, origin2, lmn2, prim_off2, exps2, coefs_raw2);
    if (!msg.empty()) TF_FAIL(ctx, TF_EINVAL, "%s", msg.c_str());
    msg = tfone::cross_overlap(ctx->bs, other, S_cross, &ctx->arena1e);
    if (!msg.empty()) TF_FAIL(ctx, TF_ENODEVICE, "%s", msg.c_str());
    return TF_OK;
}

int tf_eri_element(tf_ctx *ctx, const double *origin, const int32_t *lmn, const int32_t *prim_off, const double *exps,
                   const double *coefs_raw, double *value)
{
    if (!ctx) return TF_EINVAL;
    if (!origin || !lmn || !prim_off || !exps || !coefs_raw || !value) TF_FAIL(ctx, TF_EINVAL, "tf_eri_element: null argument");
    // A throw-away 4-AO context; (b0 b1|b2 b3) is element [0,1,2,3] of its Cartesian tensor.
    tf_ctx *tmp = tf_create(ctx->device, 0, 1);
    if (!tmp) TF_FAIL(ctx, TF_ENODEVICE, "%s", g_create_error.c_str());
    int rc = tf_set_basis(tmp, 4, origin, lmn, prim_off, exps, coefs_raw);
    if (!rc) rc = tf_build_eri(tmp, 0);
    const int32_t idx[4] = {0, 1, 2, 3};
    if (!rc) rc = tf_sample_eri(tmp, 1, idx, value);
    if (rc) ctx->err = tmp->err;
    tf_destroy(tmp);
    return rc;
}

// ---- SCF --------------------------------------------------------------------------------------------

// The parity class of every output AO of the current build, offered to the eigensolvers of a workspace (tfscf::eigh_blocked verifies on
// every call that the matrix really has the block structure, so a vector that does not fit the problem only costs the test).
static void offer_symmetry(tf_ctx *ctx, tfscf::Workspace &w, int n)
{
    static const std::vector<int> none;
    tfscf::set_symmetry(w, (ctx->have_eri && (int)ctx->hl.cls.size() == n) ? ctx->hl.cls : none);
}

int tf_orthogonaliser(tf_ctx *ctx, int n, const double *S, double *X, double *S_inv, double *smallest_eig)
{
    if (!ctx) return TF_EINVAL;
    if (n < 1 || !S || !X) TF_FAIL(ctx, TF_EINVAL, "tf_orthogonaliser: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::string msg;
    offer_symmetry(ctx, ctx->scf, n);
    int rc = tfscf::orthogonaliser(ctx->scf, n, S, X, S_inv, smallest_eig, msg);
    if (rc) ctx->err = msg;
    return rc;
}

int tf_diagonalise(tf_ctx *ctx, int n, const double *F, const double *X, double *eps, double *C)
{
    if (!ctx) return TF_EINVAL;
    if (n < 1 || !F || !X || !eps || !C) TF_FAIL(ctx, TF_EINVAL, "tf_diagonalise: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::string msg;
    offer_symmetry(ctx, ctx->scf, n);
    int rc = tfscf::diagonalise(ctx->scf, n, F, X, eps, C, msg);
    if (rc) ctx->err = msg;
    return rc;
}

// ---- Kohn-Sham exchange-correlation (next row of the hot path: SURVEY.md section 8f rank 2) --------------------------

int tf_dft_clear(tf_ctx *ctx)
{
    if (!ctx) return TF_EINVAL;
    (void)hipSetDevice(ctx->device);
    tfdft::release(ctx->grid);
    return TF_OK;
}

int tf_dft_setup(tf_ctx *ctx, int64_t n_points, const double *xyz, const double *weights, int x_functional, int c_functional, double dfx,
                 double dfc, double x_alpha)
{
    if (!ctx) return TF_EINVAL;
    if (!ctx->have_eri) TF_FAIL(ctx, TF_EINVAL, "tf_dft_setup: call tf_build_eri first (it fixes the AO representation)");
    if (n_points < 1 || !xyz || !weights || x_functional < 0 || x_functional > 3 || c_functional < 0 || c_functional > 5)
        TF_FAIL(ctx, TF_EINVAL, "tf_dft_setup: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    tfdft::release(ctx->grid);
    tfdft::Grid &g = ctx->grid;
    const tf::Basis &bs = ctx->bs;
    const int N = ctx->N;
    const long long G = n_points;
    g.G = G; g.N = N; g.xid = x_functional; g.cid = c_functional; g.dfx = dfx; g.dfc = dfc; g.x_alpha = x_alpha;
    g.gga = (x_functional >= tfdft::X_B88) || (c_functional >= tfdft::C_LYP);
    std::string err;
    tfone::DevBuf buf;
    tfone::DAO A = tfone::upload_aos(bs, buf, err);
    if (!err.empty()) TF_FAIL(ctx, TF_ENODEVICE, "%s", err.c_str());
    double *d_xyz = buf.alloc<double>((size_t)3 * G, err);
    if (!err.empty()) TF_FAIL(ctx, TF_ENOMEM, "%s", err.c_str());
    HIPCHK(ctx, hipMemcpy(d_xyz, xyz, (size_t)3 * G * sizeof(double), hipMemcpyHostToDevice));
    const size_t GN = (size_t)G * N;
    // (the grid's buffers are released by tfdft::release in tf_dft.hip.h with the runtime's own hipFree: they bypass this file's block cache)
    HIPCHK(ctx, ::hipMalloc((void **)&g.w, (size_t)G * sizeof(double)));
    HIPCHK(ctx, hipMemcpy(g.w, weights, (size_t)G * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(ctx, ::hipMalloc((void **)&g.phi, GN * sizeof(double)));
    HIPCHK(ctx, ::hipMalloc((void **)&g.dphi, (g.gga ? 3 : 1) * GN * sizeof(double)));
    HIPCHK(ctx, ::hipMalloc((void **)&g.B, GN * sizeof(double)));
    HIPCHK(ctx, ::hipMalloc((void **)&g.D, GN * sizeof(double)));
    for (double **p : {&g.rho, &g.vrho, &g.vsig, &g.ex, &g.ec}) HIPCHK(ctx, ::hipMalloc((void **)p, (size_t)G * sizeof(double)));
    HIPCHK(ctx, ::hipMalloc((void **)&g.grad, (size_t)3 * G * sizeof(double)));
    HIPCHK(ctx, ::hipMalloc((void **)&g.V, (size_t)(tfdft::VSPLIT + 2) * N * N * sizeof(double)));
    HIPCHK(ctx, ::hipMalloc((void **)&g.part, (size_t)3 * tfdft::NPART * sizeof(double)));
    tfdft::DAOs D{A.z, A.lmn, A.prim_off, A.exps, A.w};
    hipLaunchKernelGGL(tfdft::ao_on_grid_kernel, dim3((unsigned)((GN + 127) / 128)), dim3(128), 0, 0, D, d_xyz, G, N, ctx->d_csr_ptr,
                       ctx->d_csr_idx, ctx->d_csr_val, g.phi, g.dphi, g.gga ? 1 : 0);
    HIPCHK(ctx, hipDeviceSynchronize());
    HIPCHK(ctx, hipGetLastError());
    std::string msg;
    int rc = tfscf::ensure(ctx->scf, N, 6, msg);           // makes sure the rocBLAS handle exists
    if (rc) { ctx->err = msg; return rc; }
    return TF_OK;
}

int tf_dft_vxc(tf_ctx *ctx, const double *P, double *Vxc, double *n_elec, double *e_x, double *e_c)
{
    if (!ctx) return TF_EINVAL;
    if (ctx->grid.G <= 0) TF_FAIL(ctx, TF_EINVAL, "tf_dft_vxc: call tf_dft_setup first");
    if (!ctx->have_eri || ctx->grid.N != ctx->N)
        TF_FAIL(ctx, TF_EINVAL, "tf_dft_vxc: the grid was set up for a different tensor (call tf_build_eri, then tf_dft_setup again)");
    if (!P || !Vxc) TF_FAIL(ctx, TF_EINVAL, "tf_dft_vxc: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t nn = (size_t)ctx->N * ctx->N;
    HIPCHK(ctx, hipMemcpy(ctx->d_P, P, nn * sizeof(double), hipMemcpyHostToDevice));
    std::string msg;
    double o3[3];
    int rc = tfdft::vxc(ctx->scf.blas, ctx->grid, ctx->d_P, ctx->d_J, o3, msg);
    if (rc) { ctx->err = msg; return rc; }
    HIPCHK(ctx, hipMemcpy(Vxc, ctx->d_J, nn * sizeof(double), hipMemcpyDeviceToHost));
    if (n_elec) *n_elec = o3[0];
    if (e_x) *e_x = o3[1];
    if (e_c) *e_c = o3[2];
    return TF_OK;
}

// ---- AO->MO transformation and RMP2 (next row of the hot path: consumers of the resident tensor) ----------------------

static int mo_transform_device(tf_ctx *ctx, const double *C1, int n1, const double *C2, int n2, const double *C3, int n3, const double *C4,
                               int n4, double **d_out, double *seconds)
{
    const int N = ctx->N;
    std::string msg;
    int rc = tfscf::ensure(ctx->scf, N, 6, msg);
    if (rc) { ctx->err = msg; return rc; }
    if (ctx->world > 1 && !ctx->allreduce && !ctx->comm)
        TF_FAIL(ctx, TF_EINVAL, "AO->MO transformation of a sharded tensor (world > 1) needs an RCCL communicator (tf_comm_init) or the all-reduce hook (tf_set_allreduce)");
    const bool packed = ctx->layout >= 1, tiles = ctx->layout == 2;
    double *dC[4] = {nullptr, nullptr, nullptr, nullptr}, *dG[2] = {nullptr, nullptr};
    const double *hC[4] = {C1, C2, C3, C4};
    const int nk[4] = {n1, n2, n3, n4};
    const size_t total = (size_t)n1 * n2 * n3 * n4;
    // packed rows hold the pairs (kl) <= (ij): out[pq][rs] = G(C1 C2 C3 C4)[pq][rs] + G(C3 C4 C1 C2)[rs][pq]; one transformation
    // serves both terms when the two coefficient pairs are the same matrices ((ia|jb), (pq|rs) with one C)
    const bool same_pairs = n1 == n3 && n2 == n4 && std::memcmp(C1, C3, (size_t)N * n1 * sizeof(double)) == 0 &&
                            std::memcmp(C2, C4, (size_t)N * n2 * sizeof(double)) == 0;
    double secs = 0.0;
    auto cleanup = [&]() { for (int k = 0; k < 4; ++k) if (dC[k]) (void)tf_free(dC[k]); for (int k = 0; k < 2; ++k) if (dG[k]) (void)tf_free(dG[k]); };
    auto fail = [&](int code, const std::string &m) { ctx->err = m; cleanup(); if (*d_out) { (void)tf_free(*d_out); *d_out = nullptr; } return code; };
    *d_out = nullptr;
    for (int k = 0; k < 4; ++k) {
        if (tf_malloc((void **)&dC[k], (size_t)N * nk[k] * sizeof(double)) != hipSuccess) return fail(TF_ENOMEM, "AO->MO transformation: out of device memory");
        if (hipMemcpy(dC[k], hC[k], (size_t)N * nk[k] * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(TF_ENODEVICE, "AO->MO transformation: copy failed");
    }
    if (tf_malloc((void **)d_out, (total + 1) * sizeof(double)) != hipSuccess) return fail(TF_ENOMEM, "AO->MO transformation: out of device memory");   // (+ 1: the status word of the hook's exchange)
    // the short index first: a ket coefficient matrix of at most 32 columns (the occupied orbitals of (ia|jb)) goes through the hand-written
    // first quarter on the packed segments (tfmp2::mo_q1_kernel); TF_MO_Q1=0 keeps the expanded-block path (A/B, tests)
    const char *q1env = getenv("TF_MO_Q1");
    const bool q1_allowed = packed && !tiles && !(q1env && q1env[0] == '0');
    auto run = [&](int a, int b, int c, int d, double *dst) {
        double s1 = 0.0;
        if (q1_allowed && nk[c] <= 32 && nk[a] <= 32 && (size_t)N * ((nk[a] + 1) & ~1) * sizeof(double) + (size_t)N * sizeof(int) <= ((size_t)150 << 10)) {
            const size_t need = tfmp2::q1_pool_doubles(N, ctx->n_rows, nk[a], nk[b], nk[c], nk[d]) * sizeof(double);
            size_t free_b = 0, total_b = 0;
            const bool fits = need <= ctx->mo_pool_bytes || (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need + ((size_t)2 << 30) <= free_b + ctx->mo_pool_bytes + tfcache::cached());
            if (fits) {
                if (need > ctx->mo_pool_bytes) {
                    if (ctx->mo_pool) { (void)tf_free(ctx->mo_pool); ctx->mo_pool = nullptr; ctx->mo_pool_bytes = 0; }
                    if (tf_malloc((void **)&ctx->mo_pool, need) == hipSuccess) ctx->mo_pool_bytes = need;
                    else { ctx->mo_pool = nullptr; (void)hipGetLastError(); }
                }
                if (ctx->mo_pool) {
                    int r = tfmp2::transform_q1(ctx->scf.blas, ctx->d_eri, ctx->d_rowoff, ctx->d_rowsec, ctx->bl, ctx->d_row_ij, ctx->d_rowmap, ctx->n_rows, N,
                                                dC[a], nk[a], dC[b], nk[b], dC[c], nk[c], dC[d], nk[d], dst, ctx->mo_pool, &s1, msg);
                    secs += s1;
                    return r;
                }
            }
        }
        tfmp2::PackedRows pr{};
        if (packed) {
            for (int q = 0; q < 4; ++q) { pr.csize[q] = ctx->hl.csize[q]; pr.cstart[q] = ctx->hl.cstart[q]; pr.NP[q] = ctx->hl.NP[q]; }
            for (int q = 0; q < 5; ++q) pr.class_row_off[q] = ctx->class_row_off[q];
            pr.d_class_rows = ctx->d_class_rows; pr.d_row_pos = ctx->d_row_pos;
        }
        int r = tfmp2::transform(ctx->scf.blas, ctx->d_eri, ctx->d_rowmap, (packed && !tiles) ? ctx->d_rowoff : nullptr, ctx->d_rowsec, ctx->bl, pr,
                                 ctx->d_row_ij, ctx->n_rows, N, ctx->ld, dC[a], nk[a], dC[b], nk[b], dC[c], nk[c], dC[d], nk[d], dst, &s1, msg,
                                 tiles ? &ctx->tv : nullptr);
        secs += s1;
        return r;
    };
    if (!packed) {
        if ((rc = run(0, 1, 2, 3, *d_out))) return fail(rc, msg);
    } else {
        if (tf_malloc((void **)&dG[0], total * sizeof(double)) != hipSuccess) return fail(TF_ENOMEM, "AO->MO transformation: out of device memory");
        if ((rc = run(0, 1, 2, 3, dG[0]))) return fail(rc, msg);
        if (!same_pairs) {
            if (tf_malloc((void **)&dG[1], total * sizeof(double)) != hipSuccess) return fail(TF_ENOMEM, "AO->MO transformation: out of device memory");
            if ((rc = run(2, 3, 0, 1, dG[1]))) return fail(rc, msg);
        }
        const long long A = (long long)n1 * n2, B = (long long)n3 * n4;
        hipLaunchKernelGGL(add_transposed_kernel, dim3((unsigned)((B + 31) / 32), (unsigned)((A + 31) / 32)), dim3(32, 8), 0, 0, dG[0],
                           same_pairs ? dG[0] : dG[1], A, B, *d_out);
        if (hipGetLastError() != hipSuccess) return fail(TF_ENODEVICE, "AO->MO transformation: launch failed");
    }
    if (ctx->world > 1) {                                   // every rank transformed its own rows: the sum is the tensor
        if (hipDeviceSynchronize() != hipSuccess) return fail(TF_ENODEVICE, "AO->MO transformation failed on the device");
        if (ctx->comm) {
            if (comm_allreduce(ctx, *d_out, total, nullptr)) return fail(TF_ENODEVICE, ctx->err);
        } else {
            // the hook sums a buffer whose LAST element is a status word: a rank whose staging failed sends zeros and sets it
            if (hipMemset(*d_out + total, 0, sizeof(double)) != hipSuccess) return fail(TF_ENODEVICE, "AO->MO transformation: memset failed");
            const int arc = ctx->allreduce(ctx->allreduce_user, *d_out, (long long)(total + 1), nullptr);
            if (arc) return fail(TF_ENODEVICE, "AO->MO transformation: the all-reduce hook failed (code " + std::to_string(arc) + ")");
            double status = 0.0;
            if (hipMemcpy(&status, *d_out + total, sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return fail(TF_ENODEVICE, "AO->MO transformation: copy failed");
            if (status != 0.0) return fail(TF_ENODEVICE, "AO->MO transformation: the exchange step failed on a rank: every rank stops");
        }
    }
    if (hipDeviceSynchronize() != hipSuccess) return fail(TF_ENODEVICE, "AO->MO transformation failed on the device");
    cleanup();
    if (seconds) *seconds = secs;
    return TF_OK;
}

int tf_ao_to_mo(tf_ctx *ctx, int n1, const double *C1, int n2, const double *C2, int n3, const double *C3, int n4, const double *C4,
                double *out)
{
    if (!ctx) return TF_EINVAL;
    if (!ctx->have_eri) TF_FAIL(ctx, TF_EINVAL, "tf_ao_to_mo: call tf_build_eri first");
    if (n1 < 1 || n2 < 1 || n3 < 1 || n4 < 1 || !C1 || !C2 || !C3 || !C4 || !out) TF_FAIL(ctx, TF_EINVAL, "tf_ao_to_mo: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    double *d_out = nullptr;
    int rc = mo_transform_device(ctx, C1, n1, C2, n2, C3, n3, C4, n4, &d_out, nullptr);
    if (rc) return rc;
    hipError_t e = hipMemcpy(out, d_out, (size_t)n1 * n2 * n3 * n4 * sizeof(double), hipMemcpyDeviceToHost);
    (void)tf_free(d_out);
    HIPCHK(ctx, e);
    return TF_OK;
}

int tf_mp2_rhf(tf_ctx *ctx, int n_occ, int n_frozen, const double *C, const double *eps, double *e_os, double *e_ss, double *seconds)
{
    if (!ctx) return TF_EINVAL;
    if (!ctx->have_eri) TF_FAIL(ctx, TF_EINVAL, "tf_mp2_rhf: call tf_build_eri first");
    const int N = ctx->N;
    if (!C || !eps || !e_os || !e_ss || n_frozen < 0 || n_occ <= n_frozen || n_occ >= N) TF_FAIL(ctx, TF_EINVAL, "tf_mp2_rhf: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int o = n_occ - n_frozen, v = N - n_occ;
    std::vector<double> Co((size_t)N * o), Cv((size_t)N * v);
    for (int m = 0; m < N; ++m) {
        for (int i = 0; i < o; ++i) Co[(size_t)m * o + i] = C[(size_t)m * N + n_frozen + i];
        for (int a = 0; a < v; ++a) Cv[(size_t)m * v + a] = C[(size_t)m * N + n_occ + a];
    }
    double *d_g = nullptr;
    auto t0 = std::chrono::steady_clock::now();
    int rc = mo_transform_device(ctx, Co.data(), o, Cv.data(), v, Co.data(), o, Cv.data(), v, &d_g, nullptr);   // (ia|jb)
    if (rc) return rc;
    double *d_eps = nullptr, *d_part = nullptr;
    const int nblk = 1024;
    HIPCHK(ctx, tf_malloc((void **)&d_eps, (size_t)N * sizeof(double)));
    HIPCHK(ctx, tf_malloc((void **)&d_part, (size_t)2 * nblk * sizeof(double)));
    HIPCHK(ctx, hipMemcpy(d_eps, eps, (size_t)N * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(tfmp2::mp2_energy_kernel, dim3(nblk), dim3(256), 0, 0, d_g, d_eps, n_frozen, o, v, n_occ, d_part);
    std::vector<double> part(2 * nblk);
    hipError_t e = hipMemcpy(part.data(), d_part, part.size() * sizeof(double), hipMemcpyDeviceToHost);
    (void)tf_free(d_g); (void)tf_free(d_eps); (void)tf_free(d_part);
    HIPCHK(ctx, e);
    double os = 0.0, ss = 0.0;
    for (int b = 0; b < nblk; ++b) { os += part[2 * b]; ss += part[2 * b + 1]; }
    *e_os = os; *e_ss = ss;
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return TF_OK;
}

int tf_eigh_probe(tf_ctx *ctx, int n, int variant, int reps, double *seconds)
{
    if (!ctx || n < 1 || reps < 1 || !seconds) return TF_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::string msg;
    int rc = tfscf::eigh_probe(ctx->scf, n, variant, reps, seconds, msg);
    if (rc) ctx->err = msg;
    return rc;
}

int tf_jk_path_stats(tf_ctx *ctx, int64_t out[2])
{
    if (!ctx || !out) return TF_EINVAL;
    out[0] = ctx->jk_cd_passes; out[1] = ctx->jk_cd_declined;
    return TF_OK;
}

int tf_eigh_stats(tf_ctx *ctx, int64_t out[5])
{
    if (!ctx || !out) return TF_EINVAL;
    const tfscf::Workspace &w = ctx->scf;
    out[0] = w.ref_solves; out[1] = w.ref_steps; out[2] = w.ref_fallbacks; out[3] = w.sym_solves; out[4] = w.sym_declined;
    return TF_OK;
}

extern "C++" {
// A native cycle on a sharded tensor: every rank must take the same decisions from the same data.  For the duration of the cycle the
// control decisions are agreed over the ranks (agree_over_ranks) and rocBLAS runs without atomics (bitwise reproducible GEMMs; not
// otherwise: its split-K kernels need them -- the skinny GEMMs of the AO->MO transformation are ten times slower without).
struct ShardedCycleGuard {
    tf_ctx *ctx;
    bool on;
    explicit ShardedCycleGuard(tf_ctx *c) : ctx(c), on(c->world > 1 && (c->allreduce || c->comm)) {
        if (!on) { ctx->scf.agree = nullptr; return; }
        ctx->scf.agree = [c](const double *v, int nv, std::string &m) { return agree_over_ranks(c, v, nv, m); };
        if (ctx->scf.blas) (void)rocblas_set_atomics_mode(ctx->scf.blas, rocblas_atomics_not_allowed);
    }
    ~ShardedCycleGuard() {
        ctx->scf.agree = nullptr;
        if (on && ctx->scf.blas) (void)rocblas_set_atomics_mode(ctx->scf.blas, rocblas_atomics_allowed);
    }
};
}

int tf_scf_rhf(tf_ctx *ctx, const tf_scf_opts *opts, const double *S, const double *T, const double *V, const double *Fext,
               const double *X, const double *P0, double E0, int n_occ, double V_NN, tf_scf_result *out)
{
    if (!ctx) return TF_EINVAL;
    if (!ctx->have_eri) TF_FAIL(ctx, TF_EINVAL, "tf_scf_rhf: call tf_build_eri first");
    if (!opts || !S || !T || !V || !P0 || !out || n_occ < 1 || n_occ > ctx->N) TF_FAIL(ctx, TF_EINVAL, "tf_scf_rhf: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::string msg;
    auto jk = [&](const double *dP, double *dJ, double *dK, hipStream_t st) {
        const double *p[2] = {dP, dP};
        double *j[2] = {dJ, dJ}, *k[2] = {dK, dK};
        ctx->jk_try_class_diagonal = true;                          // (a cycle's densities are class-diagonal unless a field along x / y mixes the classes)
        int rcj = launch_jk(ctx, 1, p, j, k, st);
        ctx->jk_try_class_diagonal = false;
        return rcj ? rcj : allreduce_jk(ctx, 1, j, k, st);
    };
    tfscf::XCFn xc;
    if (ctx->grid.G > 0) {
        if (ctx->grid.N != ctx->N) TF_FAIL(ctx, TF_EINVAL, "tf_scf_rhf: the DFT grid was set up for a different AO dimension");
        xc = [&](const double *dP, double *dV, double *o3) { return tfdft::vxc(ctx->scf.blas, ctx->grid, dP, dV, o3, msg); };
    }
    ShardedCycleGuard guard(ctx);
    offer_symmetry(ctx, ctx->scf, ctx->N);
    int rc = tfscf::run_rhf(ctx->scf, ctx->N, *opts, S, T, V, Fext, X, P0, E0, n_occ, V_NN, jk, ctx->world, *out, msg, xc);
    if (rc && !msg.empty()) ctx->err = msg;
    return rc;
}

extern "C++" {
// Rendezvous of the cycles of a lockstep batch in their Fock builds (tf_scf_rhf_batch): every cycle runs the unmodified native cycle
// (tfscf::run_rhf) on a host thread of its own and asks for J and K of its density once per iteration; the last cycle to arrive sends
// the densities of all cycles still iterating through the tensor together (two per pass, jk_packed_kernel<2>).
struct LockstepJK {
    struct Req { const double *dP; double *dJ, *dK; };
    tf_ctx *ctx;
    std::mutex mu;
    std::condition_variable cv;
    int active = 0, waiting = 0, last_rc = TF_OK;
    long long generation = 0, passes = 0, builds = 0;
    std::vector<Req> req;
    hipEvent_t ev[2] = {nullptr, nullptr};          // "the builds of generation g are queued", alternating
    LockstepJK(tf_ctx *c, int n) : ctx(c), active(n), req((size_t)n)
    {
        (void)hipEventCreateWithFlags(&ev[0], hipEventDisableTiming);
        (void)hipEventCreateWithFlags(&ev[1], hipEventDisableTiming);
    }
    ~LockstepJK() { for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); }
    void build_locked(hipStream_t st)               // on the stream of the cycle that arrived last; the others wait for its event
    {
        int rc = TF_OK;
        const int cap = ctx->layout == 2 ? 8 : 2;   // densities per pass over the tensor: the tiles layout's wide pass takes eight, the packed kernel two
        for (int q = 0; q < waiting && rc == TF_OK; q += cap) {
            const int nd = std::min(cap, waiting - q);
            const double *p[8];
            double *j[8], *k[8];
            for (int d = 0; d < nd; ++d) { p[d] = req[q + d].dP; j[d] = req[q + d].dJ; k[d] = req[q + d].dK; }
            ctx->jk_try_class_diagonal = true;                      // (under the mutex of the batch: one pass at a time)
            rc = launch_jk(ctx, nd, p, j, k, st, nullptr);
            ctx->jk_try_class_diagonal = false;
            ++passes; builds += nd;
        }
        if (hipEventRecord(ev[generation & 1], st) != hipSuccess && rc == TF_OK) rc = TF_ENODEVICE;
        last_rc = rc;
        waiting = 0;
        ++generation;
        cv.notify_all();
    }
    int fock(const double *dP, double *dJ, double *dK, hipStream_t st)
    {
        if (hipStreamSynchronize(st) != hipSuccess) return TF_ENODEVICE;   // this cycle's density is complete before another stream reads it
        hipEvent_t done;
        int rc;
        {
            std::unique_lock<std::mutex> lk(mu);
            req[(size_t)waiting++] = Req{dP, dJ, dK};
            const long long g = generation;
            if (waiting == active) build_locked(st);
            else cv.wait(lk, [&] { return generation != g; });
            done = ev[g & 1];
            rc = last_rc;
        }
        if (hipStreamWaitEvent(st, done, 0) != hipSuccess && rc == TF_OK) rc = TF_ENODEVICE;
        return rc;
    }
    void finish(hipStream_t st)                     // a cycle has left its loop: those waiting may be complete now
    {
        std::unique_lock<std::mutex> lk(mu);
        --active;
        if (active > 0 && waiting == active) build_locked(st);
    }
};
}

int tf_scf_rhf_batch(tf_ctx *ctx, int n_cycles, const tf_scf_opts *opts, const double *S, const double *T, const double *V,
                     const double *const *Fext, const double *X, const double *const *P0, const double *E0, int n_occ, double V_NN,
                     tf_scf_result *out, int32_t *rc_out, int64_t *passes_out)
{
    if (!ctx) return TF_EINVAL;
    if (!ctx->have_eri) TF_FAIL(ctx, TF_EINVAL, "tf_scf_rhf_batch: call tf_build_eri first");
    if (n_cycles < 1 || n_cycles > 64 || !opts || !S || !T || !V || !P0 || !E0 || !out || n_occ < 1 || n_occ > ctx->N)
        TF_FAIL(ctx, TF_EINVAL, "tf_scf_rhf_batch: bad arguments");
    if (ctx->world > 1) TF_FAIL(ctx, TF_EINVAL, "tf_scf_rhf_batch: lockstep batches run on an unsharded tensor (world = 1) in this build");
    if (ctx->grid.G > 0) TF_FAIL(ctx, TF_EINVAL, "tf_scf_rhf_batch: lockstep batches run Hartree-Fock cycles (clear the DFT grid with tf_dft_clear)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    while ((int)ctx->scf_batch.size() < n_cycles) ctx->scf_batch.emplace_back(new tfscf::Workspace());
    LockstepJK ls(ctx, n_cycles);
    static const bool serial_streams = getenv("TF_BATCH_ONE_STREAM") != nullptr;      // (A/B: every cycle on the legacy default stream)
    std::vector<int> rcs((size_t)n_cycles, TF_OK);
    std::vector<std::string> msgs((size_t)n_cycles);
    std::vector<std::thread> threads;
    // At most TF_BATCH_STREAMS (4) streams for the cycles, shared round-robin: the O(N^3) steps of four cycles side by side fill the chip;
    // more of them at once only slow each other down.  (A stream per cycle was the same thing while the process had 4 hardware queues; with
    // the 16 the tensor build wants, eight truly concurrent cycles took 0.94 s for the N = 400 polarisability instead of 0.58 s.)
    static const int batch_streams = getenv("TF_BATCH_STREAMS") ? std::max(1, atoi(getenv("TF_BATCH_STREAMS"))) : 4;
    std::vector<hipStream_t> pool((size_t)std::min(n_cycles, batch_streams), nullptr);
    if (!serial_streams)
        for (auto &ps : pool)
            if (hipStreamCreateWithFlags(&ps, hipStreamNonBlocking) != hipSuccess) ps = nullptr;
    for (int c = 0; c < n_cycles; ++c)
        threads.emplace_back([&, c] {
            // a host thread per cycle on one of the pool's non-blocking streams: the O(N^3) steps of different cycles overlap on the device
            (void)hipSetDevice(ctx->device);
            hipStream_t st = serial_streams ? nullptr : pool[(size_t)c % pool.size()];
            tfscf::t_stream = st;
            auto jk = [&](const double *dP, double *dJ, double *dK, hipStream_t s2) { return ls.fock(dP, dJ, dK, s2); };
            offer_symmetry(ctx, *ctx->scf_batch[(size_t)c], ctx->N);
            rcs[(size_t)c] = tfscf::run_rhf(*ctx->scf_batch[(size_t)c], ctx->N, *opts, S, T, V, Fext ? Fext[c] : nullptr, X, P0[c], E0[c], n_occ, V_NN, jk, 1,
                                             out[c], msgs[(size_t)c], tfscf::XCFn());
            (void)hipStreamSynchronize(st);
            ls.finish(st);
            (void)hipStreamSynchronize(st);
            tfscf::t_stream = nullptr;
        });
    for (auto &t : threads) t.join();
    for (hipStream_t ps : pool)
        if (ps) (void)hipStreamDestroy(ps);
    int rc = TF_OK;
    for (int c = 0; c < n_cycles; ++c) {
        if (rc_out) rc_out[c] = rcs[(size_t)c];
        if (rcs[(size_t)c] != TF_OK && rc == TF_OK) { rc = rcs[(size_t)c]; ctx->err = "cycle " + std::to_string(c) + ": " + msgs[(size_t)c]; }
    }
    if (passes_out) { passes_out[0] = ls.passes; passes_out[1] = ls.builds; }
    return rc;
}

int tf_scf_uhf(tf_ctx *ctx, const tf_scf_opts *opts, const double *S, const double *T, const double *V, const double *Fext,
               const double *X, const double *P0_alpha, const double *P0_beta, double E0, int n_alpha, int n_beta, double V_NN,
               tf_scf_uhf_result *out)
{
    if (!ctx) return TF_EINVAL;
    if (!ctx->have_eri) TF_FAIL(ctx, TF_EINVAL, "tf_scf_uhf: call tf_build_eri first");
    if (!opts || !S || !T || !V || !P0_alpha || !P0_beta || !out || n_alpha < 1 || n_alpha > ctx->N || n_beta < 0 || n_beta > n_alpha)
        TF_FAIL(ctx, TF_EINVAL, "tf_scf_uhf: bad arguments");
    if (ctx->grid.G > 0) TF_FAIL(ctx, TF_EINVAL, "tf_scf_uhf: unrestricted Kohn-Sham is not implemented (call tf_dft_clear)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::string msg;
    auto jk2 = [&](const double *dPa, const double *dPb, double *dJa, double *dJb, double *dKa, double *dKb, hipStream_t st) {
        const double *p[2] = {dPa, dPb};
        double *j[2] = {dJa, dJb}, *k[2] = {dKa, dKb};
        ctx->jk_try_class_diagonal = true;
        int rcj = launch_jk(ctx, 2, p, j, k, st);
        ctx->jk_try_class_diagonal = false;
        return rcj ? rcj : allreduce_jk(ctx, 2, j, k, st);
    };
    tfscf::UhfOut uo;
    for (int sp = 0; sp < 2; ++sp) { uo.P[sp] = out->P_spin[sp]; uo.C[sp] = out->C_spin[sp]; uo.eps[sp] = out->eps_spin[sp]; uo.F[sp] = out->F_spin[sp]; }
    ShardedCycleGuard guard(ctx);
    offer_symmetry(ctx, ctx->scf, ctx->N);
    int rc = tfscf::run_uhf(ctx->scf, ctx->N, *opts, S, T, V, Fext, X, P0_alpha, P0_beta, E0, n_alpha, n_beta, V_NN, jk2, ctx->world,
                            out->common, uo, msg);
    if (rc && !msg.empty()) ctx->err = msg;
    return rc;
}

}  // extern "C"
