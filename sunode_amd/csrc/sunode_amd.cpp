/*
 * libsunode_amd.so -- host side of the C ABI in include/sunode_amd.h.
 *
 * Thin by design: loads the per-problem gfx950 code object (hipModuleLoad), owns the device
 * buffers (trajectory arena, staging copies for host-memory calls, tolerance vector) and
 * launches the forward / backward integrator kernels of bdf_kernels.hip on one HIP stream,
 * timing them with HIP events.  No integrator arithmetic happens on the host and there is no
 * CPU fallback: every entry point fails with SA_ERR_HIP if no GPU / code object is usable.
 */
#include <hip/hip_runtime.h>

#include <algorithm>

#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sunode_amd.h"
#include "sa_device_abi.h"

static_assert(SA_DEVICE_ABI_VERSION == SA_ABI_VERSION, "sa_device_abi.h and include/sunode_amd.h disagree on the device ABI");

static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(SA_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                               \
    } while (0)

/* Select a device for the duration of a scope and give the calling thread its previous one back: construction,
   destruction and queries run on the USER's thread (a torch / PyMC process whose own allocations follow the
   current device); the compute entry points, documented to do so, leave their handle's device selected. */
struct DeviceScope {
    int prev = -1;
    bool ok = false;
    explicit DeviceScope(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return SA_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes < 256 ? 256 : bytes;
        HIP_TRY(hipMalloc(&p, want));
        cap = want;
        return SA_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct Guard;

struct sa_solver {
    std::string path;              /* the code object this handle runs */
    Guard *guard = nullptr;        /* differential guard (sa_solver_attach_guard); shadows have none */
    int device = 0;
    int n = 0, p = 0, r = 0;
    int group = 1;                 /* lanes per instance: 1 = thread-per-instance, 2^k = lane group / workgroup build */
    int64_t ws_doubles = 0;        /* per-instance workspace (0: register builds; small: lane groups; the whole state: memory-resident) */
    bool point_major = false;      /* arena records laid out [point][..][instance] (memory-resident kernel) instead of [instance][point] */
    int32_t rec_doubles = 0;       /* doubles per arena record: 8 + 6n (table records) unless the code object says otherwise */
    DevBuf ws;
    hipModule_t module = nullptr;
    hipFunction_t k_forward = nullptr, k_backward = nullptr, k_eval = nullptr, k_math = nullptr;
    hipFunction_t k_sens = nullptr;            /* only in forward-sensitivity builds */
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   /* fwd start/stop, bwd start/stop */
    bool have_fwd_time = false, have_bwd_time = false;
    sa_options opt{};
    std::vector<double> atol;
    DevBuf d_atol;
    std::vector<double> constraints;
    DevBuf d_constraints;
    bool have_constraints = false;
    /* trajectory arena + forward bookkeeping of the last forward batch (see "trajectory arena" below) */
    DevBuf traj, traj_np, fwd_status;
    int64_t traj_stride = 0;
    int32_t traj_rows = 0;         /* rows the resident arena was launched with */
    int32_t fwd_B = 0;
    double fwd_t0 = 0.0;
    bool tiled = false;            /* the trajectories are NOT resident: the backward call re-integrates tile by tile */
    std::vector<int32_t> h_np;     /* tiled mode: points per instance, from the counting pass */
    std::vector<int32_t> h_scratch;
    DevBuf keep_y0, keep_tvals;    /* tiled mode: the forward call's initial states and output grid */
    int32_t fwd_n_t = 0;
    DevBuf t_yout, t_status, t_stats, t_np;   /* outputs of the re-integration (discarded: identical to the forward call's) */
    int32_t rows_hint = 0;         /* largest per-instance point count seen so far on this handle */
    /* the forward call only enqueues; whether its trajectories are resident is settled at the start of the backward
       call (resolve_forward) from the device-side overflow word */
    bool pending = false;          /* a forward batch was launched and its arena outcome has not been read yet */
    bool resident_attempt = false; /* ... it was launched with arena writes (rows = traj_rows) */
    DevBuf d_overflow;             /* int32: max point count of the instances that outgrew traj_rows (0: none) */
    size_t budget = 0;             /* arena budget of the last forward call (the backward call must use the same) */
    std::vector<int32_t> full_idx; /* instances whose 64-instance group exceeds the budget: backward status ARENA_FULL */
    int64_t stat_tiles = 0, stat_arena_bytes = 0;
    /* staging for SA_MEM_HOST calls */
    DevBuf s_y0, s_ps, s_pr, s_tvals, s_yout, s_status, s_stats, s_grads, s_gout, s_lout;
    DevBuf s_misc[12];
};

static int launch(sa_solver *s, hipFunction_t f, int32_t n_items, void *args, size_t args_size, int lanes_per_item = 1)
{
    /* up to 64 lanes per item: 64-thread blocks carrying 64/lanes items; above: one item per block */
    const int block = lanes_per_item > 64 ? lanes_per_item : 64;
    const int per_block = lanes_per_item > 64 ? 1 : 64 / lanes_per_item;
    unsigned grid = (unsigned)((n_items + per_block - 1) / per_block);
    if (grid == 0) return SA_OK;
    void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &args_size,
                      HIP_LAUNCH_PARAM_END};
    HIP_TRY(hipModuleLaunchKernel(f, grid, 1, 1, block, 1, 1, 0, s->stream, nullptr, config));
    return SA_OK;
}

static int apply_options(sa_solver *s, const sa_options *opt)
{
    if (!opt || opt->struct_size != (int32_t)sizeof(sa_options))
        return fail(SA_ERR_ARG, "sa_options.struct_size mismatch (got %d, want %zu)",
                    opt ? opt->struct_size : -1, sizeof(sa_options));
    if (s->n > 0 && !opt->atol) return fail(SA_ERR_ARG, "sa_options.atol is NULL");
    if (opt->traj_capacity < 2) return fail(SA_ERR_ARG, "traj_capacity must be >= 2");
    if (opt->arena_bytes < 0) return fail(SA_ERR_ARG, "arena_bytes must be >= 0");
    s->opt = *opt;
    s->atol.assign(opt->atol, opt->atol + s->n);
    s->opt.atol = nullptr;
    s->have_constraints = (opt->constraints != nullptr && s->n > 0);
    s->opt.constraints = nullptr;
    if (s->have_constraints) {
        s->constraints.assign(opt->constraints, opt->constraints + s->n);
        for (double c : s->constraints)
            if (!(c == 0.0 || c == 1.0 || c == -1.0 || c == 2.0 || c == -2.0))
                return fail(SA_ERR_ARG, "constraints entries must be 0, +-1 or +-2");
        int rc2 = s->d_constraints.ensure(sizeof(double) * s->n);
        if (rc2) return rc2;
        HIP_TRY(hipMemcpy(s->d_constraints.p, s->constraints.data(), sizeof(double) * s->n, hipMemcpyHostToDevice));
    }
    int rc = s->d_atol.ensure(sizeof(double) * (s->n > 0 ? s->n : 1));
    if (rc) return rc;
    if (s->n > 0)
        HIP_TRY(hipMemcpy(s->d_atol.p, s->atol.data(), sizeof(double) * s->n, hipMemcpyHostToDevice));
    return SA_OK;
}

extern "C" int sa_abi_version(void) { return SA_ABI_VERSION; }
extern "C" const char *sa_last_error(void) { return g_err.c_str(); }

extern "C" int sa_device_count(int32_t *count)
{
    if (!count) return fail(SA_ERR_ARG, "null argument");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    *count = ndev;
    return SA_OK;
}

extern "C" int sa_device_memory(int32_t device, int64_t *free_bytes, int64_t *total_bytes)
{
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(SA_ERR_ARG, "device %d out of range (%d visible)", device, ndev);
    DeviceScope scope(device);     /* a query must not move the caller's current device */
    if (!scope.ok) return fail(SA_ERR_HIP, "hipSetDevice(%d) failed", device);
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    if (free_bytes) *free_bytes = (int64_t)free_b;
    if (total_bytes) *total_bytes = (int64_t)total_b;
    return SA_OK;
}

extern "C" int sa_solver_create(const char *path, const sa_options *opt, sa_solver **out)
{
    if (!path || !opt || !out) return fail(SA_ERR_ARG, "null argument");
    *out = nullptr;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (opt->device < 0 || opt->device >= ndev)
        return fail(SA_ERR_ARG, "device %d out of range (%d visible)", opt->device, ndev);
    DeviceScope scope(opt->device);
    if (!scope.ok) return fail(SA_ERR_HIP, "hipSetDevice(%d) failed", opt->device);
    sa_solver *s = new sa_solver();
    s->device = opt->device;
    s->path = path;
    hipError_t e = hipModuleLoad(&s->module, path);
    if (e != hipSuccess) {
        delete s;
        return fail(SA_ERR_MODULE, "hipModuleLoad(%s) failed: %s", path, hipGetErrorString(e));
    }
    hipDeviceptr_t meta_p = nullptr;
    size_t meta_sz = 0;
    int32_t meta[6] = {0, 0, 0, 0, 0, 0};
    e = hipModuleGetGlobal(&meta_p, &meta_sz, s->module, "sa_meta");
    if (e != hipSuccess || meta_sz != sizeof(meta) ||
        hipMemcpyDtoH(meta, meta_p, sizeof(meta)) != hipSuccess || meta[3] != SA_ABI_VERSION ||
        meta[4] < 1 || meta[4] > 1024 || (meta[4] & (meta[4] - 1)) != 0 || meta[5] < 0) {
        (void)hipModuleUnload(s->module);
        delete s;
        return fail(SA_ERR_MODULE, "%s: sa_meta missing or ABI mismatch", path);
    }
    s->n = meta[0]; s->p = meta[1]; s->r = meta[2]; s->group = meta[4]; s->ws_doubles = meta[5];
    s->rec_doubles = 8 + 6 * s->n;
    {   /* compact-trajectory builds store {order, t, y[n]} per step and say so */
        hipDeviceptr_t rp = nullptr;
        size_t rsz = 0;
        int32_t rec = 0;
        if (hipModuleGetGlobal(&rp, &rsz, s->module, "sa_traj_rec") == hipSuccess && rsz == sizeof(rec) &&
            hipMemcpyDtoH(&rec, rp, sizeof(rec)) == hipSuccess && rec >= 2) s->rec_doubles = rec;
        else (void)hipGetLastError();
    }
    {   /* arena layout: every instance's records contiguous ([instance][point]: the backward pass walks an
           instance's points in order) for every kernel family EXCEPT the memory-resident one, which announces its
           [point][field][instance] layout.  (Until round 4 the choice hung on ws_doubles == 0, which the lane-group /
           workgroup builds -- they have a small workspace for the saved Jacobian -- do not satisfy: they got the
           point-major layout the comments said they did not have.)  SA_TRAJ_LAYOUT=point|instance: tuning override. */
        hipDeviceptr_t rp = nullptr;
        size_t rsz = 0;
        s->point_major = hipModuleGetGlobal(&rp, &rsz, s->module, "sa_traj_point_major") == hipSuccess;
        if (!s->point_major) (void)hipGetLastError();
        const char *force = getenv("SA_TRAJ_LAYOUT");
        if (force && !s->point_major) s->point_major = (force[0] == 'p');
    }
    const char *names[4] = {"sa_k_forward", "sa_k_backward", "sa_k_eval", "sa_k_math"};
    hipFunction_t *slots[4] = {&s->k_forward, &s->k_backward, &s->k_eval, &s->k_math};
    for (int i = 0; i < 4; i++) {
        e = hipModuleGetFunction(slots[i], s->module, names[i]);
        if (e != hipSuccess) {
            (void)hipModuleUnload(s->module);
            delete s;
            return fail(SA_ERR_MODULE, "%s: kernel %s not found", path, names[i]);
        }
    }
    if (hipModuleGetFunction(&s->k_sens, s->module, "sa_k_sens") != hipSuccess) {
        s->k_sens = nullptr;
        (void)hipGetLastError();     /* the optional kernel is absent: do not leave hipErrorNotFound behind for
                                        other users of the runtime in this process (PyTorch checks it) */
    }
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) {
        (void)hipModuleUnload(s->module);
        delete s;
        return fail(SA_ERR_HIP, "hipStreamCreate failed");
    }
    s->own_stream = true;
    for (int i = 0; i < 4; i++) (void)hipEventCreate(&s->ev[i]);
    int rc = apply_options(s, opt);
    if (rc) { sa_solver_destroy(s); return rc; }
    *out = s;
    return SA_OK;
}

static void guard_free(sa_solver *s);
static int guard_set_options(sa_solver *s, const sa_options *opt);

extern "C" void sa_solver_destroy(sa_solver *s)
{
    if (!s) return;
    DeviceScope scope(s->device);
    guard_free(s);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    DevBuf *bufs[] = {&s->d_atol, &s->traj, &s->traj_np, &s->fwd_status, &s->keep_y0, &s->keep_tvals,
                      &s->t_yout, &s->t_status, &s->t_stats, &s->t_np, &s->s_y0,
                      &s->s_ps, &s->s_pr, &s->s_tvals, &s->s_yout, &s->s_status, &s->s_stats, &s->s_grads,
                      &s->s_gout, &s->s_lout, &s->ws, &s->d_constraints, &s->d_overflow};
    for (DevBuf *b : bufs) b->release();
    for (DevBuf &b : s->s_misc) b.release();
    for (int i = 0; i < 4; i++) if (s->ev[i]) (void)hipEventDestroy(s->ev[i]);
    if (s->own_stream && s->stream) (void)hipStreamDestroy(s->stream);
    if (s->module) (void)hipModuleUnload(s->module);
    delete s;
}

extern "C" int sa_solver_set_options(sa_solver *s, const sa_options *opt)
{
    if (!s) return fail(SA_ERR_ARG, "null solver");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    sa_options o = *opt;
    o.device = s->device;
    int rc = apply_options(s, &o);
    if (rc) return rc;
    return guard_set_options(s, &o);
}

extern "C" int sa_solver_sizes(const sa_solver *s, int32_t *n, int32_t *p, int32_t *r)
{
    if (!s) return fail(SA_ERR_ARG, "null solver");
    if (n) *n = s->n;
    if (p) *p = s->p;
    if (r) *r = s->r;
    return SA_OK;
}

extern "C" int sa_set_stream(sa_solver *s, void *stream)
{
    if (!s) return fail(SA_ERR_ARG, "null solver");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->own_stream) { (void)hipStreamDestroy(s->stream); s->own_stream = false; }
    if (stream) {
        s->stream = (hipStream_t)stream;
    } else {
        HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
        s->own_stream = true;
    }
    return SA_OK;
}

extern "C" int sa_synchronize(sa_solver *s)
{
    if (!s) return fail(SA_ERR_ARG, "null solver");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return SA_OK;
}

static int resolve_forward(sa_solver *s);

extern "C" int sa_arena_info(sa_solver *s, int64_t *arena_bytes, int64_t *tiles, int32_t *tiled)
{
    if (!s) return fail(SA_ERR_ARG, "null solver");
    HIP_TRY(hipSetDevice(s->device));
    {
        int rc0 = resolve_forward(s);
        if (rc0) return rc0;
    }
    if (arena_bytes) *arena_bytes = s->stat_arena_bytes;
    if (tiles) *tiles = s->stat_tiles;
    if (tiled) *tiled = s->tiled ? 1 : 0;
    return SA_OK;
}

extern "C" int sa_last_kernel_ms(sa_solver *s, float *fwd, float *bwd)
{
    if (!s) return fail(SA_ERR_ARG, "null solver");
    HIP_TRY(hipSetDevice(s->device));
    if (fwd) {
        *fwd = -1.0f;
        if (s->have_fwd_time) { HIP_TRY(hipEventSynchronize(s->ev[1])); HIP_TRY(hipEventElapsedTime(fwd, s->ev[0], s->ev[1])); }
    }
    if (bwd) {
        *bwd = -1.0f;
        if (s->have_bwd_time) { HIP_TRY(hipEventSynchronize(s->ev[3])); HIP_TRY(hipEventElapsedTime(bwd, s->ev[2], s->ev[3])); }
    }
    return SA_OK;
}

/* stage a host array into a device buffer (async on the solver stream) */
static int stage_in(sa_solver *s, DevBuf &b, const void *host, size_t bytes, const void **dev)
{
    int rc = b.ensure(bytes ? bytes : 8);
    if (rc) return rc;
    if (bytes) HIP_TRY(hipMemcpyAsync(b.p, host, bytes, hipMemcpyHostToDevice, s->stream));
    *dev = b.p;
    return SA_OK;
}

/* memory-resident builds: one [ws_doubles][stride] block, reused by forward and backward */
static int bind_workspace(sa_solver *s, int32_t B, double **ws, int64_t *stride)
{
    *ws = nullptr;
    *stride = 0;
    if (s->ws_doubles == 0) return SA_OK;
    const int64_t st = ((int64_t)B + 63) / 64 * 64;
    int rc = s->ws.ensure(sizeof(double) * (size_t)s->ws_doubles * (size_t)st);
    if (rc) return rc;
    *ws = (double *)s->ws.p;
    *stride = st;
    return SA_OK;
}

/* ---- trajectory arena ---------------------------------------------------------------------------
 * CVODES keeps the adjoint "data points" of CVodeF in host memory, check point by check point, and
 * re-integrates the forward problem segment by segment when the backward pass needs points it no longer
 * holds (sunode: CVodeAdjInit(checkpoint_n = 500 000), solver.py:533,588 -- in effect unbounded).  Here the
 * points of a whole batch live in ONE device arena traj[rows][stride][8+6n] and the same two regimes exist:
 *
 *  resident   rows x roundup64(B) records fit the budget (sa_options.arena_bytes, default 96 GiB): the forward call stores
 *             every step, the backward call reads them.  rows starts at 512 and follows the largest point
 *             count seen on the handle (x1.25), never more than sa_options.traj_capacity.
 *  tiled      otherwise, or when an instance ran out of rows (kernel status SA_TRAJ_FULL): the forward call
 *             runs the identical integration WITHOUT arena writes (SA_MODE_ADJ_COUNT: y_out / status / stats
 *             as usual, plus the number of points per instance); the backward call then walks the batch in
 *             contiguous tiles of 64-instance groups sized so that tile_instances x max_points x record fits
 *             the budget, re-integrates each tile forward with storage (bit-identical: same kernel, same
 *             inputs) and runs the adjoint on it -- CVODES' check-point re-integration with one check point
 *             at t0.  Cost: one extra forward pass; memory: the budget, whatever B and the step counts are.
 * An instance that needs more rows than traj_capacity, or a 64-instance group that alone exceeds the budget,
 * is reported with status SA_STATUS_ARENA_FULL (never as CV_TOO_MUCH_WORK, which the reference reserves for
 * the mxstep x retries budget of one CVode call).
 */
static const int32_t SA_FIRST_ROWS = 512;

static int64_t round64(int64_t v) { return (v + 63) / 64 * 64; }

static size_t arena_budget(const sa_solver *s)
{
    if (s->opt.arena_bytes > 0) return (size_t)s->opt.arena_bytes;
    size_t free_b = 0, total_b = 0;
    size_t dflt = (size_t)96 << 30;                            /* default: 96 GiB, a third of the 288 GB ... */
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {      /* ... but at most 60 % of what is free (+ what we hold) */
        size_t avail = (size_t)(0.6 * (double)(free_b + s->traj.cap));
        if (avail < dflt) dflt = avail;
    }
    return dflt;
}

static size_t record_bytes(const sa_solver *s) { return sizeof(double) * (size_t)s->rec_doubles; }

struct FwdLaunch {
    int mode; int32_t B, n_t, rem_stride, rows; int64_t stride; double t0;
    const double *y0, *ps, *pr, *tvals; double *y_out; int32_t *status; int64_t *stats; int32_t *traj_np;
};

static int launch_forward(sa_solver *s, const FwdLaunch &f)
{
    sa_fwd_args a;
    memset(&a, 0, sizeof a);
    a.B = f.B; a.n_t = f.n_t; a.mode = f.mode; a.mxstep = s->opt.mxstep; a.max_retries = s->opt.max_retries_fwd;
    a.traj_cap = f.rows; a.rem_stride = f.rem_stride;
    a.traj_max = s->opt.traj_capacity; a.overflow = (int32_t *)s->d_overflow.p;
    a.t0 = f.t0; a.rtol = s->opt.rtol; a.atol = (const double *)s->d_atol.p;
    a.y0 = f.y0; a.ps = f.ps; a.pr = f.pr; a.tvals = f.tvals; a.y_out = f.y_out; a.status = f.status; a.stats = f.stats;
    a.constraints = s->have_constraints ? (const double *)s->d_constraints.p : nullptr;
    /* arena layout: the register / lane-group / workgroup kernels keep every instance's records contiguous
       ([instance][point]: the backward pass walks an instance's points in order, so consecutive records share cache
       lines and DRAM pages); the memory-resident kernel keeps its [point][field][instance] layout */
    if (!s->point_major) { a.traj_istride = f.rows; a.traj_stride = 1; }
    else { a.traj_istride = 1; a.traj_stride = f.stride; }
    a.traj = (double *)s->traj.p; a.traj_np = f.traj_np;
    int rc;
    if ((rc = bind_workspace(s, f.B, &a.ws, &a.ws_stride))) return rc;
    return launch(s, s->k_forward, f.B, &a, sizeof a, s->group);
}

static int guard_forward(sa_solver *s, int mode, int32_t B, const double *d_y0, const double *d_ps, const double *d_pr,
                         int32_t rem_stride, double t0, const double *d_tv, int32_t n_t, const int32_t *d_status,
                         const int64_t *d_stats);
static bool guard_wants(const sa_solver *s, uint32_t kind);
static bool guard_recheck_due(sa_solver *s, uint32_t kind, int32_t B, const int32_t *status, const int64_t *stats);
static bool guard_switched(sa_solver *s);

static int forward_common(sa_solver *s, int mode, int mem, int32_t B, const double *y0, const double *ps,
                          const double *pr, int32_t rem_stride, double t0, const double *tvals, int32_t n_t,
                          double *y_out, int32_t *status, int64_t *stats)
{
    if (!s) return fail(SA_ERR_ARG, "null solver");
    if (B < 0 || n_t < 0) return fail(SA_ERR_ARG, "negative size");
    if (rem_stride != 0 && rem_stride != s->r) return fail(SA_ERR_ARG, "rem_stride must be 0 or n_rem=%d", s->r);
    HIP_TRY(hipSetDevice(s->device));
    if (B == 0 || n_t == 0) return SA_OK;
    const size_t nB = (size_t)B;
    const double *d_y0 = y0, *d_ps = ps, *d_pr = pr, *d_tv = tvals;
    double *d_yout = y_out;
    int32_t *d_status = status;
    int64_t *d_stats = stats;
    int rc;
    if (mem == SA_MEM_HOST) {
        const void *q;
        if ((rc = stage_in(s, s->s_y0, y0, sizeof(double) * nB * s->n, &q))) return rc; d_y0 = (const double *)q;
        if ((rc = stage_in(s, s->s_ps, ps, sizeof(double) * nB * s->p, &q))) return rc; d_ps = (const double *)q;
        if ((rc = stage_in(s, s->s_pr, pr, sizeof(double) * (rem_stride ? nB : 1) * s->r, &q))) return rc; d_pr = (const double *)q;
        if ((rc = stage_in(s, s->s_tvals, tvals, sizeof(double) * n_t, &q))) return rc; d_tv = (const double *)q;
        if ((rc = s->s_yout.ensure(sizeof(double) * nB * n_t * s->n))) return rc; d_yout = (double *)s->s_yout.p;
        if ((rc = s->s_status.ensure(sizeof(int32_t) * nB))) return rc; d_status = (int32_t *)s->s_status.p;
        if ((rc = s->s_stats.ensure(sizeof(int64_t) * nB * SA_N_STATS))) return rc; d_stats = (int64_t *)s->s_stats.p;
    } else if (mem != SA_MEM_DEVICE) {
        return fail(SA_ERR_ARG, "mem must be SA_MEM_HOST or SA_MEM_DEVICE");
    }
    const uint32_t gkind = mode == SA_MODE_PLAIN ? SA_GUARD_PLAIN : SA_GUARD_ADJOINT;
    FwdLaunch f{mode, B, n_t, rem_stride, 2, 0, t0, d_y0, d_ps, d_pr, d_tv, d_yout, d_status, d_stats, nullptr};
    if (mode == SA_MODE_PLAIN) {
        HIP_TRY(hipEventRecord(s->ev[0], s->stream));
        if ((rc = launch_forward(s, f))) return rc;
        HIP_TRY(hipEventRecord(s->ev[1], s->stream));
        s->have_fwd_time = true;
    } else {
        /* Enqueue only (no host synchronisation: SA_MEM_DEVICE callers overlap this with their own work).  Resident
           attempt when the rows this handle expects fit the budget, otherwise the pass that only counts; either way
           the kernel delivers y_out / status / stats and the per-instance point counts, and raises *overflow when an
           instance outgrew the rows.  The backward call reads that word and decides (resolve_forward). */
        const int64_t stride = round64(B);
        const size_t rec = record_bytes(s), budget = arena_budget(s);
        s->budget = budget;
        const int64_t fit_rows = (int64_t)(budget / ((size_t)stride * rec));
        int64_t want = s->rows_hint > 0 ? (int64_t)(1.25 * s->rows_hint) + 8 : SA_FIRST_ROWS;
        if (want < SA_FIRST_ROWS) want = SA_FIRST_ROWS;
        if (want > s->opt.traj_capacity) want = s->opt.traj_capacity;
        if ((rc = s->traj_np.ensure(sizeof(int32_t) * stride))) return rc;
        if ((rc = s->fwd_status.ensure(sizeof(int32_t) * stride))) return rc;
        if ((rc = s->d_overflow.ensure(sizeof(int32_t)))) return rc;
        if ((rc = s->keep_y0.ensure(sizeof(double) * nB * (size_t)(s->n > 0 ? s->n : 1)))) return rc;
        if ((rc = s->keep_tvals.ensure(sizeof(double) * (size_t)n_t))) return rc;
        f.traj_np = (int32_t *)s->traj_np.p;
        HIP_TRY(hipMemsetAsync(s->d_overflow.p, 0, sizeof(int32_t), s->stream));
        HIP_TRY(hipEventRecord(s->ev[0], s->stream));
        s->resident_attempt = (want <= fit_rows);
        if (s->resident_attempt) {
            if ((rc = s->traj.ensure((size_t)want * (size_t)stride * rec))) return rc;
            f.mode = SA_MODE_ADJ_FWD; f.rows = (int32_t)want; f.stride = stride;
            s->traj_stride = stride; s->traj_rows = (int32_t)want;
        } else {
            f.mode = SA_MODE_ADJ_COUNT; f.rows = 2; f.stride = stride;
        }
        if ((rc = launch_forward(s, f))) return rc;
        HIP_TRY(hipEventRecord(s->ev[1], s->stream));
        s->have_fwd_time = true;
        /* what a re-integration needs (the caller may reuse its buffers after this call) */
        if (s->keep_y0.p != (const void *)d_y0)       /* (the guard's repeat of a forward pass reads them in place) */
            HIP_TRY(hipMemcpyAsync(s->keep_y0.p, d_y0, sizeof(double) * nB * s->n, hipMemcpyDeviceToDevice, s->stream));
        if (s->keep_tvals.p != (const void *)d_tv)
            HIP_TRY(hipMemcpyAsync(s->keep_tvals.p, d_tv, sizeof(double) * (size_t)n_t, hipMemcpyDeviceToDevice, s->stream));
        HIP_TRY(hipMemcpyAsync(s->fwd_status.p, d_status, sizeof(int32_t) * nB, hipMemcpyDeviceToDevice, s->stream));
        s->pending = true;
        s->tiled = !s->resident_attempt;
        s->fwd_B = B;
        s->fwd_t0 = t0;
        s->fwd_n_t = n_t;
    }
    if (mem == SA_MEM_HOST) {
        HIP_TRY(hipMemcpyAsync(y_out, d_yout, sizeof(double) * nB * n_t * s->n, hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipMemcpyAsync(status, d_status, sizeof(int32_t) * nB, hipMemcpyDeviceToHost, s->stream));
        if (stats)
            HIP_TRY(hipMemcpyAsync(stats, d_stats, sizeof(int64_t) * nB * SA_N_STATS, hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
    }
    /* differential guard ("differential guard" below): the first batch of this kind on a code object nobody has
       compared yet -- or a later host-memory batch that shows a status code / failure regime the verified sample never
       had -- goes through the default AND the conservative build: a sample chosen from the statuses and counters the
       main launch has just produced, compared bit for bit */
    if (guard_wants(s, gkind) || (mem == SA_MEM_HOST && guard_recheck_due(s, gkind, B, status, stats))) {
        if ((rc = guard_forward(s, mode, B, d_y0, d_ps, d_pr, rem_stride, t0, d_tv, n_t, d_status, d_stats))) return rc;
        if (guard_switched(s))      /* the builds differ: THIS batch again, with the conservative code object */
            return forward_common(s, mode, mem, B, y0, ps, pr, rem_stride, t0, tvals, n_t, y_out, status, stats);
    }
    return SA_OK;
}

extern "C" int sa_solve_batch(sa_solver *s, int mem, int32_t B, const double *y0, const double *ps,
                              const double *pr, int32_t rem_stride, double t0, const double *tvals, int32_t n_t,
                              double *y_out, int32_t *status, int64_t *stats)
{
    return forward_common(s, SA_MODE_PLAIN, mem, B, y0, ps, pr, rem_stride, t0, tvals, n_t, y_out, status, stats);
}

static int guard_sens(sa_solver *s, int ism, const double *scaling, int32_t B, const double *d_y0, const double *d_ps,
                      const double *d_pr, int32_t rem_stride, const double *d_s0, double t0, const double *d_tv,
                      int32_t n_t, const int32_t *d_status, const int64_t *d_stats);

extern "C" int sa_solve_sens_batch(sa_solver *s, int mem, int ism, const double *scaling, int32_t B,
                                   const double *y0, const double *ps, const double *pr, int32_t rem_stride,
                                   const double *sens0, double t0, const double *tvals, int32_t n_t,
                                   double *y_out, double *sens_out, int32_t *status, int64_t *stats)
{
    if (!s) return fail(SA_ERR_ARG, "null solver");
    if (!s->k_sens) return fail(SA_ERR_ARG, "this code object was built without forward-sensitivity support");
    if (B < 0 || n_t < 0) return fail(SA_ERR_ARG, "negative size");
    if (ism != 0 && ism != 1) return fail(SA_ERR_ARG, "ism must be 0 (simultaneous) or 1 (staggered)");
    if (rem_stride != 0 && rem_stride != s->r) return fail(SA_ERR_ARG, "rem_stride must be 0 or n_rem=%d", s->r);
    HIP_TRY(hipSetDevice(s->device));
    if (B == 0 || n_t == 0) return SA_OK;
    const size_t nB = (size_t)B, np_n = (size_t)s->p * s->n;
    std::vector<double> pbar((size_t)(s->p > 0 ? s->p : 1), 1.0);
    if (scaling) for (int i = 0; i < s->p; i++) pbar[i] = scaling[i] < 0 ? -scaling[i] : scaling[i];
    const double *d_y0 = y0, *d_ps = ps, *d_pr = pr, *d_tv = tvals, *d_s0 = sens0;
    double *d_yout = y_out, *d_sout = sens_out;
    int32_t *d_status = status;
    int64_t *d_stats = stats;
    int rc;
    const void *q;
    if ((rc = stage_in(s, s->s_misc[0], pbar.data(), sizeof(double) * pbar.size(), &q))) return rc;
    const double *d_pbar = (const double *)q;
    if (mem == SA_MEM_HOST) {
        if ((rc = stage_in(s, s->s_y0, y0, sizeof(double) * nB * s->n, &q))) return rc; d_y0 = (const double *)q;
        if ((rc = stage_in(s, s->s_ps, ps, sizeof(double) * nB * s->p, &q))) return rc; d_ps = (const double *)q;
        if ((rc = stage_in(s, s->s_pr, pr, sizeof(double) * (rem_stride ? nB : 1) * s->r, &q))) return rc; d_pr = (const double *)q;
        if ((rc = stage_in(s, s->s_tvals, tvals, sizeof(double) * n_t, &q))) return rc; d_tv = (const double *)q;
        if ((rc = stage_in(s, s->s_grads, sens0, sizeof(double) * nB * np_n, &q))) return rc; d_s0 = (const double *)q;
        if ((rc = s->s_yout.ensure(sizeof(double) * nB * n_t * s->n))) return rc; d_yout = (double *)s->s_yout.p;
        if ((rc = s->s_gout.ensure(sizeof(double) * nB * n_t * (np_n ? np_n : 1)))) return rc; d_sout = (double *)s->s_gout.p;
        if ((rc = s->s_status.ensure(sizeof(int32_t) * nB))) return rc; d_status = (int32_t *)s->s_status.p;
        if ((rc = s->s_stats.ensure(sizeof(int64_t) * nB * SA_N_STATS))) return rc; d_stats = (int64_t *)s->s_stats.p;
    } else if (mem != SA_MEM_DEVICE) {
        return fail(SA_ERR_ARG, "mem must be SA_MEM_HOST or SA_MEM_DEVICE");
    }
    sa_sens_args a;
    memset(&a, 0, sizeof a);
    a.B = B; a.n_t = n_t; a.ism = ism; a.mxstep = s->opt.mxstep; a.max_retries = s->opt.max_retries_fwd;
    a.rem_stride = rem_stride; a.t0 = t0; a.rtol = s->opt.rtol;
    a.atol = (const double *)s->d_atol.p; a.pbar = d_pbar;
    a.y0 = d_y0; a.ps = d_ps; a.pr = d_pr; a.sens0 = d_s0; a.tvals = d_tv;
    a.y_out = d_yout; a.sens_out = d_sout; a.status = d_status; a.stats = d_stats;
    if ((rc = bind_workspace(s, B, &a.ws, &a.ws_stride))) return rc;
    HIP_TRY(hipEventRecord(s->ev[0], s->stream));
    if ((rc = launch(s, s->k_sens, B, &a, sizeof a, s->group))) return rc;
    HIP_TRY(hipEventRecord(s->ev[1], s->stream));
    s->have_fwd_time = true;
    if (mem == SA_MEM_HOST) {
        HIP_TRY(hipMemcpyAsync(y_out, d_yout, sizeof(double) * nB * n_t * s->n, hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipMemcpyAsync(sens_out, d_sout, sizeof(double) * nB * n_t * np_n, hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipMemcpyAsync(status, d_status, sizeof(int32_t) * nB, hipMemcpyDeviceToHost, s->stream));
        if (stats)
            HIP_TRY(hipMemcpyAsync(stats, d_stats, sizeof(int64_t) * nB * SA_N_STATS, hipMemcpyDeviceToHost, s->stream));
    }
    HIP_TRY(hipStreamSynchronize(s->stream));      /* pbar staging buffer is host-owned */
    if (guard_wants(s, SA_GUARD_SENS) || (mem == SA_MEM_HOST && guard_recheck_due(s, SA_GUARD_SENS, B, status, stats))) {
        if ((rc = guard_sens(s, ism, scaling, B, d_y0, d_ps, d_pr, rem_stride, d_s0, t0, d_tv, n_t, d_status, d_stats)))
            return rc;
        if (guard_switched(s))      /* the builds differ: THIS batch again, with the conservative code object */
            return sa_solve_sens_batch(s, mem, ism, scaling, B, y0, ps, pr, rem_stride, sens0, t0, tvals, n_t, y_out,
                                       sens_out, status, stats);
    }
    return SA_OK;
}

extern "C" int sa_solve_forward_batch(sa_solver *s, int mem, int32_t B, const double *y0, const double *ps,
                                      const double *pr, int32_t rem_stride, double t0, const double *tvals,
                                      int32_t n_t, double *y_out, int32_t *status, int64_t *stats)
{
    return forward_common(s, SA_MODE_ADJ_FWD, mem, B, y0, ps, pr, rem_stride, t0, tvals, n_t, y_out, status, stats);
}

/* Settle the arena outcome of the last forward batch (one 4-byte read-back; the only host synchronisation of a
   device-memory forward + backward pair, and it sits in the backward call).  Resident and nothing overflowed: the
   backward kernel reads the arena.  Otherwise fetch the point counts and switch to tiled re-integration; a
   64-instance group that cannot fit the budget even alone is taken out (backward status SA_STATUS_ARENA_FULL). */
static int resolve_forward(sa_solver *s)
{
    if (!s->pending) return SA_OK;
    s->pending = false;
    const size_t nB = (size_t)s->fwd_B;
    const size_t rec = record_bytes(s);
    s->full_idx.clear();
    if (s->resident_attempt) {
        int32_t ov = 0;
        HIP_TRY(hipMemcpyAsync(&ov, s->d_overflow.p, sizeof(int32_t), hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
        if (ov == 0) {
            s->tiled = false;
            s->stat_arena_bytes = (int64_t)((size_t)s->traj_rows * (size_t)s->traj_stride * rec);
            return SA_OK;
        }
    }
    s->tiled = true;
    s->h_np.resize(nB);
    s->h_scratch.resize(nB);
    HIP_TRY(hipMemcpyAsync(s->h_np.data(), s->traj_np.p, sizeof(int32_t) * nB, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipMemcpyAsync(s->h_scratch.data(), s->fwd_status.p, sizeof(int32_t) * nB, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    const int64_t group_rows = (int64_t)(s->budget / ((size_t)64 * rec));
    int32_t seen = 0;
    for (size_t i = 0; i < nB; i++) {
        if (s->h_np[i] > seen) seen = s->h_np[i];
        if (s->h_np[i] > group_rows) {              /* (more than traj_capacity points: already failed in the kernel) */
            s->h_np[i] = 0;
            s->h_scratch[i] = SA_STATUS_ARENA_FULL;
            s->full_idx.push_back((int32_t)i);
        }
    }
    if (!s->full_idx.empty())                       /* the library-owned copy: the backward kernel answers CV_NO_FWD */
        HIP_TRY(hipMemcpyAsync(s->fwd_status.p, s->h_scratch.data(), sizeof(int32_t) * nB, hipMemcpyHostToDevice, s->stream));
    if (seen > s->rows_hint) s->rows_hint = seen;
    return SA_OK;
}

static int guard_backward(sa_solver *s, int32_t B, const double *d_ps, const double *d_pr, int32_t rem_stride,
                          double t0, double tend, const double *d_tv, int32_t n_t, const double *d_g,
                          int64_t grads_stride);
static bool guard_backward_due(const sa_solver *s);

extern "C" int sa_solve_backward_batch(sa_solver *s, int mem, int32_t B, const double *ps, const double *pr,
                                       int32_t rem_stride, double t0, double tend, const double *tvals,
                                       int32_t n_t, const double *grads, int64_t grads_stride, double *grad_out,
                                       double *lamda_out, int32_t *status, int64_t *stats)
{
    return sa_solve_backward_batch_all(s, mem, B, ps, pr, rem_stride, t0, tend, tvals, n_t, grads, grads_stride,
                                       grad_out, lamda_out, nullptr, nullptr, status, stats);
}

extern "C" int sa_solve_backward_batch_all(sa_solver *s, int mem, int32_t B, const double *ps, const double *pr,
                                           int32_t rem_stride, double t0, double tend, const double *tvals,
                                           int32_t n_t, const double *grads, int64_t grads_stride,
                                           double *grad_out, double *lamda_out, double *lamda_all_out,
                                           double *quad_all_out, int32_t *status, int64_t *stats)
{
    if (!s) return fail(SA_ERR_ARG, "null solver");
    if (B != s->fwd_B) return fail(SA_ERR_ARG, "backward batch %d does not match the last forward batch %d", B, s->fwd_B);
    if (rem_stride != 0 && rem_stride != s->r) return fail(SA_ERR_ARG, "rem_stride must be 0 or n_rem=%d", s->r);
    if (grads_stride != 0 && grads_stride != (int64_t)n_t * s->n) return fail(SA_ERR_ARG, "grads_stride must be 0 or n_t*n");
    HIP_TRY(hipSetDevice(s->device));
    if (B == 0) return SA_OK;
    {
        int rc0 = resolve_forward(s);
        if (rc0) return rc0;
    }
    const size_t nB = (size_t)B;
    const double *d_ps = ps, *d_pr = pr, *d_tv = tvals, *d_g = grads;
    double *d_gout = grad_out, *d_lout = lamda_out, *d_lall = lamda_all_out, *d_qall = quad_all_out;
    int32_t *d_status = status;
    int64_t *d_stats = stats;
    int rc;
    if (mem == SA_MEM_HOST) {
        const void *q;
        if ((rc = stage_in(s, s->s_ps, ps, sizeof(double) * nB * s->p, &q))) return rc; d_ps = (const double *)q;
        if ((rc = stage_in(s, s->s_pr, pr, sizeof(double) * (rem_stride ? nB : 1) * s->r, &q))) return rc; d_pr = (const double *)q;
        if ((rc = stage_in(s, s->s_tvals, tvals, sizeof(double) * n_t, &q))) return rc; d_tv = (const double *)q;
        if ((rc = stage_in(s, s->s_grads, grads, sizeof(double) * (grads_stride ? nB : 1) * n_t * s->n, &q))) return rc; d_g = (const double *)q;
        if ((rc = s->s_gout.ensure(sizeof(double) * nB * (s->p > 0 ? s->p : 1)))) return rc; d_gout = (double *)s->s_gout.p;
        if ((rc = s->s_lout.ensure(sizeof(double) * nB * (s->n > 0 ? s->n : 1)))) return rc; d_lout = (double *)s->s_lout.p;
        if (lamda_all_out) { if ((rc = s->s_misc[1].ensure(sizeof(double) * nB * n_t * (s->n > 0 ? s->n : 1)))) return rc; d_lall = (double *)s->s_misc[1].p; }
        if (quad_all_out) { if ((rc = s->s_misc[2].ensure(sizeof(double) * nB * n_t * (s->p > 0 ? s->p : 1)))) return rc; d_qall = (double *)s->s_misc[2].p; }
        if ((rc = s->s_status.ensure(sizeof(int32_t) * nB))) return rc; d_status = (int32_t *)s->s_status.p;
        if ((rc = s->s_stats.ensure(sizeof(int64_t) * nB * SA_N_STATS))) return rc; d_stats = (int64_t *)s->s_stats.p;
    } else if (mem != SA_MEM_DEVICE) {
        return fail(SA_ERR_ARG, "mem must be SA_MEM_HOST or SA_MEM_DEVICE");
    }
    if (guard_backward_due(s)) {
        if ((rc = guard_backward(s, B, d_ps, d_pr, rem_stride, t0, tend, d_tv, n_t, d_g, grads_stride))) return rc;
    }
    sa_bwd_args a;
    memset(&a, 0, sizeof a);
    a.n_t = n_t; a.mxstep = s->opt.mxstep; a.max_retries = s->opt.max_retries_bwd;
    a.rem_stride = rem_stride; a.grads_stride = grads_stride;
    a.t0 = t0; a.tend = tend; a.tinitial = s->fwd_t0;
    a.rtolB = s->opt.rtolB; a.atolB = s->opt.atolB; a.rtolQB = s->opt.rtolQB; a.atolQB = s->opt.atolQB;
    a.tvals = d_tv;
    HIP_TRY(hipEventRecord(s->ev[2], s->stream));
    if (!s->tiled) {
        a.B = B; a.traj_cap = s->traj_rows;
        if (!s->point_major) { a.traj_istride = s->traj_rows; a.traj_stride = 1; }
        else { a.traj_istride = 1; a.traj_stride = s->traj_stride; }
        a.ps = d_ps; a.pr = d_pr; a.grads = d_g; a.grad_out = d_gout; a.lamda_out = d_lout;
        a.status = d_status; a.fwd_status = (const int32_t *)s->fwd_status.p; a.stats = d_stats;
        a.traj = (const double *)s->traj.p; a.traj_np = (const int32_t *)s->traj_np.p;
        a.lamda_all = d_lall; a.quad_all = d_qall;
        if ((rc = bind_workspace(s, B, &a.ws, &a.ws_stride))) return rc;
        if ((rc = launch(s, s->k_backward, B, &a, sizeof a, s->group))) return rc;
    } else {
        /* tiled: re-integrate the forward problem tile by tile with exactly sized storage, adjoint per tile */
        const size_t rec = record_bytes(s), budget = s->budget;      /* the forward call's (not re-evaluated) */
        const size_t np_ = (size_t)s->p, nn = (size_t)s->n;
        /* tile boundaries: as few tiles as the budget allows (greedy over 64-instance groups), then balanced --
           equal-sized tiles keep every launch wide enough to fill the chip -- as long as each still fits */
        const int64_t n_groups = (B + 63) / 64;
        std::vector<int32_t> gmax((size_t)n_groups, 2);            /* largest point count of every 64-instance group */
        for (int64_t i = 0; i < B; i++)
            if (s->h_np[(size_t)i] > gmax[(size_t)(i / 64)]) gmax[(size_t)(i / 64)] = s->h_np[(size_t)i];
        auto rows_of = [&](int64_t lo_, int64_t hi_) {             /* lo_, hi_ on group boundaries (hi_ may be B) */
            int64_t r2 = 2;
            for (int64_t g = lo_ / 64; g < (hi_ + 63) / 64; g++) if (gmax[(size_t)g] > r2) r2 = gmax[(size_t)g];
            return r2;
        };
        auto fits = [&](int64_t lo_, int64_t hi_, int64_t *rows_out) {
            const int64_t r2 = rows_of(lo_, hi_);
            if (rows_out) *rows_out = r2;
            return (size_t)round64(hi_ - lo_) * (size_t)r2 * rec <= budget;
        };
        std::vector<int64_t> cuts;                  /* greedy, one pass with a running maximum */
        for (int64_t lo_ = 0; lo_ < B;) {
            int64_t hi_ = (lo_ + 64 < B) ? lo_ + 64 : B;
            int64_t r2 = gmax[(size_t)(lo_ / 64)];
            while (hi_ < B) {
                const int64_t nhi = (hi_ + 64 < B) ? hi_ + 64 : B;
                const int64_t rn = gmax[(size_t)(hi_ / 64)] > r2 ? gmax[(size_t)(hi_ / 64)] : r2;
                if ((size_t)round64(nhi - lo_) * (size_t)rn * rec > budget) break;
                hi_ = nhi; r2 = rn;
            }
            cuts.push_back(hi_);
            lo_ = hi_;
        }
        if (cuts.size() > 1) {                      /* balanced alternative with the same number of tiles */
            const int64_t per = round64((B + (int64_t)cuts.size() - 1) / (int64_t)cuts.size());
            std::vector<int64_t> even;
            bool ok = true;
            for (int64_t lo_ = 0; lo_ < B && ok; lo_ += per) {
                const int64_t hi_ = (lo_ + per < B) ? lo_ + per : B;
                ok = fits(lo_, hi_, nullptr);
                even.push_back(hi_);
            }
            if (ok && even.size() <= cuts.size()) cuts.swap(even);
        }
        int64_t lo = 0;
        for (size_t ti = 0; ti < cuts.size(); ti++) {
            const int64_t hi = cuts[ti];
            int64_t rows = 2;
            (void)fits(lo, hi, &rows);
            const int32_t tB = (int32_t)(hi - lo);
            const int64_t stride = round64(tB);
            {   /* a batch that turns out to fit as ONE tile will be resident from the next call on: allocate the rows
                   that call is going to ask for right away (tens of GB are not freed and re-allocated twice) */
                int64_t alloc_rows = rows;
                const int64_t next = (int64_t)(1.25 * s->rows_hint) + 8;
                if (cuts.size() == 1 && next > rows && next <= s->opt.traj_capacity &&
                    (size_t)next * (size_t)stride * rec <= budget) alloc_rows = next;
                if ((rc = s->traj.ensure((size_t)alloc_rows * (size_t)stride * rec))) return rc;
            }
            if ((rc = s->t_yout.ensure(sizeof(double) * (size_t)tB * (size_t)s->fwd_n_t * (nn ? nn : 1)))) return rc;
            if ((rc = s->t_status.ensure(sizeof(int32_t) * (size_t)stride))) return rc;
            if ((rc = s->t_stats.ensure(sizeof(int64_t) * (size_t)stride * SA_N_STATS))) return rc;
            if ((rc = s->t_np.ensure(sizeof(int32_t) * (size_t)stride))) return rc;
            FwdLaunch f{SA_MODE_ADJ_FWD, tB, s->fwd_n_t, rem_stride, (int32_t)rows, stride, s->fwd_t0,
                        (const double *)s->keep_y0.p + (size_t)lo * nn, d_ps + (size_t)lo * np_,
                        d_pr + (size_t)lo * (size_t)rem_stride, (const double *)s->keep_tvals.p,
                        (double *)s->t_yout.p, (int32_t *)s->t_status.p, (int64_t *)s->t_stats.p, (int32_t *)s->t_np.p};
            if ((rc = launch_forward(s, f))) return rc;
            a.B = tB; a.traj_cap = (int32_t)rows;
            if (!s->point_major) { a.traj_istride = (int32_t)rows; a.traj_stride = 1; }
            else { a.traj_istride = 1; a.traj_stride = stride; }
            a.ps = d_ps + (size_t)lo * np_; a.pr = d_pr + (size_t)lo * (size_t)rem_stride;
            a.grads = d_g + (size_t)lo * (size_t)grads_stride;
            a.grad_out = d_gout + (size_t)lo * np_; a.lamda_out = d_lout + (size_t)lo * nn;
            a.status = d_status + lo; a.fwd_status = (const int32_t *)s->fwd_status.p + lo;
            a.stats = d_stats + (size_t)lo * SA_N_STATS;
            a.traj = (const double *)s->traj.p; a.traj_np = (const int32_t *)s->t_np.p;
            a.lamda_all = d_lall ? d_lall + (size_t)lo * (size_t)n_t * nn : nullptr;
            a.quad_all = d_qall ? d_qall + (size_t)lo * (size_t)n_t * np_ : nullptr;
            if ((rc = bind_workspace(s, tB, &a.ws, &a.ws_stride))) return rc;
            if ((rc = launch(s, s->k_backward, tB, &a, sizeof a, s->group))) return rc;
            s->stat_tiles++;
            if ((int64_t)((size_t)rows * stride * rec) > s->stat_arena_bytes) s->stat_arena_bytes = (int64_t)((size_t)rows * stride * rec);
            lo = hi;
        }
    }
    HIP_TRY(hipEventRecord(s->ev[3], s->stream));
    s->have_bwd_time = true;
    if (!s->full_idx.empty()) {     /* instances taken out by resolve_forward: say why (their gradients are NaN already) */
        static const int32_t full = SA_STATUS_ARENA_FULL;
        for (int32_t i : s->full_idx)
            HIP_TRY(hipMemcpyAsync(d_status + i, &full, sizeof(int32_t), hipMemcpyHostToDevice, s->stream));
    }
    if (mem == SA_MEM_HOST) {
        HIP_TRY(hipMemcpyAsync(grad_out, d_gout, sizeof(double) * nB * s->p, hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipMemcpyAsync(lamda_out, d_lout, sizeof(double) * nB * s->n, hipMemcpyDeviceToHost, s->stream));
        if (lamda_all_out)
            HIP_TRY(hipMemcpyAsync(lamda_all_out, d_lall, sizeof(double) * nB * n_t * s->n, hipMemcpyDeviceToHost, s->stream));
        if (quad_all_out)
            HIP_TRY(hipMemcpyAsync(quad_all_out, d_qall, sizeof(double) * nB * n_t * s->p, hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipMemcpyAsync(status, d_status, sizeof(int32_t) * nB, hipMemcpyDeviceToHost, s->stream));
        if (stats)
            HIP_TRY(hipMemcpyAsync(stats, d_stats, sizeof(int64_t) * nB * SA_N_STATS, hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
    }
    return SA_OK;
}


/* ---- differential guard ----------------------------------------------------------------------------
 * include/sunode_amd.h, sa_solver_attach_guard.  Two shadow handles (the default and the conservative code object)
 * integrate a SAMPLE of the caller's batch into guard-owned output buffers; the host compares the bytes.
 *
 * The sample (<= n_sample instances) is chosen AFTER the main launch of the batch, from its own statuses and counters
 * (guard_select): the first 16 instances, one instance per failure code, the instances with the most steps, error-test
 * failures, convergence failures, linear set-ups, Jacobian evaluations and retries, and an even stride over the rest
 * of the batch -- a miscompiled block that only rarely-taken paths reach (order reduction after repeated error-test
 * failures, the recoverable-rhs retry, the general-pivot LU) is compared if ANY instance of the batch takes it, not
 * only if one of the first 64 does.  The sample's rows are gathered into guard-owned device arrays (a prefix of the
 * batch is used in place).  While a kind is verified, host-memory calls (whose statuses / counters are on the host
 * anyway) are scanned for a status code or a failure-counter regime the verified sample never showed; the first such
 * batch is checked again, once per kind, within the first SA_GUARD_RECHECK_CALLS calls.
 * The shadows are created when a check is due and destroyed when nothing is pending any more.
 */
struct GuardSeen {                  /* what the verified sample of a kind covered */
    std::vector<int32_t> codes;
    int64_t max_nst = 0, max_netf = 0, max_ncfn = 0;
    bool valid = false;
};
struct Guard {
    std::string safe_path;
    sa_solver *fast = nullptr, *safe = nullptr;
    int32_t want = 64;
    uint32_t pending = 0, verified = 0, differs = 0;
    int32_t checks[3] = {0, 0, 0}, best[3] = {0, 0, 0};
    bool using_safe = false;
    bool just_switched = false;    /* set by guard_switch, consumed by the entry point that repeats its batch */
    int32_t adj_k = 0;             /* adjoint check in flight: the shadows hold the forward pass of this many instances */
    int32_t fwd_only = 0;          /* adjoint-kind forward checks that no backward call completed */
    bool wait_backward = false;    /* ... after two of them: no further forward check until a backward call was seen */
    std::vector<int32_t> idx;      /* the sample of the batch under check (ascending instance indices) */
    bool prefix = true;            /* idx == 0..k-1: the caller's arrays are used in place */
    DevBuf in[5];                  /* gathered rows of the sample: y0, ps, pr, sens0, grads */
    GuardSeen seen[3];
    int32_t recheck_left[3] = {1, 1, 1}, window[3] = {0, 0, 0};
    std::vector<int32_t> h_status;
    std::vector<int64_t> h_stats;
    DevBuf out[2][4];
    std::vector<unsigned char> host[2];
    std::string detail;
};
static const int32_t SA_GUARD_MIN_SAMPLE = 16, SA_GUARD_MAX_SMALL_CHECKS = 3, SA_GUARD_RECHECK_CALLS = 32,
                     SA_GUARD_MAX_FWD_ONLY = 2;

static int kind_slot(uint32_t kind) { return kind == SA_GUARD_PLAIN ? 0 : kind == SA_GUARD_ADJOINT ? 1 : 2; }

static bool guard_wants(const sa_solver *s, uint32_t kind) { return s->guard && (s->guard->pending & kind); }

static bool guard_backward_due(const sa_solver *s)
{
    return s->guard && ((s->guard->pending & SA_GUARD_ADJOINT) || s->guard->wait_backward);
}

static void guard_drop_shadows(Guard *g)
{
    if (g->fast) sa_solver_destroy(g->fast);
    if (g->safe) sa_solver_destroy(g->safe);
    g->fast = g->safe = nullptr;
    for (auto &row : g->out) for (DevBuf &b : row) b.release();
    for (DevBuf &b : g->in) b.release();
    g->adj_k = 0;
}

static void guard_free(sa_solver *s)
{
    if (!s->guard) return;
    guard_drop_shadows(s->guard);
    delete s->guard;
    s->guard = nullptr;
}

static sa_options shadow_options(const sa_solver *s)
{
    sa_options o = s->opt;
    o.struct_size = (int32_t)sizeof(sa_options);
    o.device = s->device;
    o.atol = s->atol.data();
    o.constraints = s->have_constraints ? s->constraints.data() : nullptr;
    return o;
}

static int guard_set_options(sa_solver *s, const sa_options *)
{
    Guard *g = s->guard;
    if (!g) return SA_OK;
    sa_options o = shadow_options(s);
    for (sa_solver *sh : {g->fast, g->safe})
        if (sh) { int rc = apply_options(sh, &o); if (rc) return rc; }
    return SA_OK;
}

/* the shadow handles, created when the first check is due (not at attach: a handle whose kinds are all verified, or
   that never runs the kind still pending, never pays for two extra modules, streams and buffer sets) */
static int guard_shadows(sa_solver *s)
{
    Guard *g = s->guard;
    sa_options o = shadow_options(s);
    int rc;
    if (!g->fast && (rc = sa_solver_create(s->path.c_str(), &o, &g->fast))) return rc;
    if (!g->safe && (rc = sa_solver_create(g->safe_path.c_str(), &o, &g->safe))) return rc;
    for (sa_solver *sh : {g->fast, g->safe})
        if (sh->n != s->n || sh->p != s->p || sh->r != s->r || sh->group != s->group || sh->ws_doubles != s->ws_doubles ||
            sh->rec_doubles != s->rec_doubles || sh->point_major != s->point_major || (sh->k_sens != nullptr) != (s->k_sens != nullptr))
            return fail(SA_ERR_MODULE, "guard: %s is not a build of the same source and options as %s",
                        sh->path.c_str(), s->path.c_str());
    return SA_OK;
}

/* the handle runs the conservative code object from now on */
static void guard_switch(sa_solver *s)
{
    Guard *g = s->guard;
    if (g->using_safe || !g->safe) return;
    std::swap(s->module, g->safe->module);
    std::swap(s->k_forward, g->safe->k_forward);
    std::swap(s->k_backward, g->safe->k_backward);
    std::swap(s->k_eval, g->safe->k_eval);
    std::swap(s->k_math, g->safe->k_math);
    std::swap(s->k_sens, g->safe->k_sens);
    std::swap(s->path, g->safe->path);
    g->using_safe = true;
    g->just_switched = true;
}

static bool guard_switched(sa_solver *s)
{
    if (!s->guard || !s->guard->just_switched) return false;
    s->guard->just_switched = false;
    return true;
}

/* ---- the sample ---- */
static void guard_take(std::vector<char> &in, std::vector<int32_t> &idx, int32_t i, int32_t k)
{
    if ((int32_t)idx.size() < k && !in[(size_t)i]) { in[(size_t)i] = 1; idx.push_back(i); }
}

/* the `count` instances with the largest positive stats[slot] (ties: lowest index) */
static void guard_take_top(std::vector<char> &in, std::vector<int32_t> &idx, int32_t B, const int64_t *stats, int slot,
                           int count, int32_t k, bool positive_only)
{
    for (int c = 0; c < count; c++) {
        int32_t arg = -1;
        int64_t bestv = positive_only ? 0 : -1;
        for (int32_t i = 0; i < B; i++) {
            const int64_t v = stats[(size_t)i * SA_N_STATS + slot];
            if (!in[(size_t)i] && v > bestv) { bestv = v; arg = i; }
        }
        if (arg < 0) return;
        guard_take(in, idx, arg, k);
    }
}

static void guard_select(Guard *g, int32_t B, const int32_t *status, const int64_t *stats)
{
    const int32_t k = B < g->want ? B : g->want;
    g->idx.clear();
    std::vector<char> in((size_t)B, 0);
    for (int32_t i = 0; i < B && i < 16; i++) guard_take(in, g->idx, i, k);
    if (k < B && status) {                       /* one instance per failure code, then more failures (<= 8) */
        std::vector<int32_t> codes;
        int taken = 0;
        for (int32_t i = 0; i < B && taken < 8; i++)
            if (status[i] != 0 && std::find(codes.begin(), codes.end(), status[i]) == codes.end()) {
                codes.push_back(status[i]);
                if (!in[(size_t)i]) { guard_take(in, g->idx, i, k); taken++; }
            }
        for (int32_t i = 0; i < B && taken < 8; i++)
            if (status[i] != 0 && !in[(size_t)i]) { guard_take(in, g->idx, i, k); taken++; }
    }
    if (k < B && stats) {
        guard_take_top(in, g->idx, B, stats, SA_ST_NST, 4, k, false);
        guard_take_top(in, g->idx, B, stats, SA_ST_NETF, 8, k, true);
        guard_take_top(in, g->idx, B, stats, SA_ST_NCFN, 8, k, true);
        guard_take_top(in, g->idx, B, stats, SA_ST_NSETUPS, 4, k, false);
        guard_take_top(in, g->idx, B, stats, SA_ST_NJE, 4, k, false);
        guard_take_top(in, g->idx, B, stats, SA_ST_RETRIES, 4, k, true);
    }
    const int32_t rest = k - (int32_t)g->idx.size();     /* an even stride over the whole batch */
    for (int32_t j = 0; j < rest; j++) guard_take(in, g->idx, (int32_t)(((int64_t)2 * j + 1) * B / ((int64_t)2 * rest)), k);
    for (int32_t i = 0; i < B && (int32_t)g->idx.size() < k; i++) guard_take(in, g->idx, i, k);
    std::sort(g->idx.begin(), g->idx.end());
    g->prefix = true;
    for (size_t j = 0; j < g->idx.size(); j++) if (g->idx[j] != (int32_t)j) g->prefix = false;
}

/* rows idx[] of a [B][row_bytes] device array -> buf (runs of consecutive rows in one copy); a prefix sample, a shared
   array (per_instance false) or a NULL array is used in place */
static int guard_gather(sa_solver *s, DevBuf &buf, const void *src, size_t row_bytes, bool per_instance, const void **out)
{
    Guard *g = s->guard;
    *out = src;
    if (!src || !per_instance || g->prefix || !row_bytes) return SA_OK;
    int rc;
    if ((rc = buf.ensure(row_bytes * g->idx.size()))) return rc;
    for (size_t j = 0; j < g->idx.size();) {
        size_t e = j + 1;
        while (e < g->idx.size() && g->idx[e] == g->idx[e - 1] + 1) e++;
        HIP_TRY(hipMemcpyAsync((char *)buf.p + j * row_bytes, (const char *)src + (size_t)g->idx[j] * row_bytes,
                               (e - j) * row_bytes, hipMemcpyDeviceToDevice, s->stream));
        j = e;
    }
    *out = buf.p;
    return SA_OK;
}

/* statuses / counters of the finished main launch on the host, the sample chosen from them */
static int guard_sample(sa_solver *s, int32_t B, const int32_t *d_status, const int64_t *d_stats)
{
    Guard *g = s->guard;
    g->h_status.clear();
    g->h_stats.clear();
    HIP_TRY(hipStreamSynchronize(s->stream));            /* the main launch (and the staged inputs) are complete */
    const bool need = B > g->want;
    if (need && d_status) {
        g->h_status.resize((size_t)B);
        HIP_TRY(hipMemcpy(g->h_status.data(), d_status, sizeof(int32_t) * (size_t)B, hipMemcpyDeviceToHost));
    }
    if (need && d_stats) {
        g->h_stats.resize((size_t)B * SA_N_STATS);
        HIP_TRY(hipMemcpy(g->h_stats.data(), d_stats, sizeof(int64_t) * (size_t)B * SA_N_STATS, hipMemcpyDeviceToHost));
    }
    guard_select(g, B, g->h_status.empty() ? nullptr : g->h_status.data(), g->h_stats.empty() ? nullptr : g->h_stats.data());
    return SA_OK;
}

struct GuardBuf { const char *name; size_t row_bytes; size_t cmp_bytes; };   /* per instance: stored / compared */

/* bring the outputs of both shadows to the host and compare; *same = verdict */
static int guard_compare(sa_solver *s, const char *what, int32_t k, const GuardBuf *bufs, int nbufs, bool *same)
{
    Guard *g = s->guard;
    *same = true;
    for (int b = 0; b < nbufs; b++) {
        const size_t bytes = bufs[b].row_bytes * (size_t)k;
        if (!bytes) continue;
        sa_solver *sh[2] = {g->fast, g->safe};
        for (int w = 0; w < 2; w++) {
            g->host[w].resize(bytes);
            HIP_TRY(hipMemcpyAsync(g->host[w].data(), g->out[w][b].p, bytes, hipMemcpyDeviceToHost, sh[w]->stream));
            HIP_TRY(hipStreamSynchronize(sh[w]->stream));
        }
        for (int32_t i = 0; i < k && *same; i++) {
            const unsigned char *x = g->host[0].data() + (size_t)i * bufs[b].row_bytes;
            const unsigned char *y = g->host[1].data() + (size_t)i * bufs[b].row_bytes;
            if (memcmp(x, y, bufs[b].cmp_bytes) != 0) {
                size_t off = 0;
                while (off < bufs[b].cmp_bytes && x[off] == y[off]) off++;
                char msg[512];
                snprintf(msg, sizeof msg, "%s: %s of instance %d differs between the default build %s and the "
                         "conservative build %s (element %zu)", what, bufs[b].name,
                         (size_t)i < g->idx.size() ? g->idx[(size_t)i] : i, g->fast->path.c_str(),
                         g->safe->path.c_str(), off / 8);
                g->detail = msg;
                *same = false;
            }
        }
        if (!*same) break;
    }
    return SA_OK;
}

/* what the sample of a passed check covered (shadow `fast` outputs are still in g->host[0] only for the last buffer:
   re-read status and counters of the default shadow) */
static int guard_remember(sa_solver *s, uint32_t kind, int32_t k, int status_buf, int stats_buf)
{
    Guard *g = s->guard;
    GuardSeen &seen = g->seen[kind_slot(kind)];
    std::vector<int32_t> st((size_t)k);
    std::vector<int64_t> ct((size_t)k * SA_N_STATS);
    HIP_TRY(hipMemcpy(st.data(), g->out[0][status_buf].p, sizeof(int32_t) * (size_t)k, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ct.data(), g->out[0][stats_buf].p, sizeof(int64_t) * (size_t)k * SA_N_STATS, hipMemcpyDeviceToHost));
    for (int32_t i = 0; i < k; i++) {
        if (std::find(seen.codes.begin(), seen.codes.end(), st[(size_t)i]) == seen.codes.end()) seen.codes.push_back(st[(size_t)i]);
        const int64_t *c = ct.data() + (size_t)i * SA_N_STATS;
        if (c[SA_ST_NST] > seen.max_nst) seen.max_nst = c[SA_ST_NST];
        if (c[SA_ST_NETF] > seen.max_netf) seen.max_netf = c[SA_ST_NETF];
        if (c[SA_ST_NCFN] > seen.max_ncfn) seen.max_ncfn = c[SA_ST_NCFN];
    }
    seen.valid = true;
    return SA_OK;
}

/* book a finished check of `kind` over k instances */
static void guard_book(sa_solver *s, uint32_t kind, int32_t k, bool same)
{
    Guard *g = s->guard;
    const int slot = kind_slot(kind);
    g->checks[slot]++;
    if (k > g->best[slot]) g->best[slot] = k;
    if (!same) {
        g->differs |= kind;
        g->verified &= ~kind;
        g->pending = 0;                       /* conservative code object from here on: nothing left to compare */
        for (int i = 0; i < 3; i++) g->recheck_left[i] = 0;
        guard_switch(s);
    } else if (k >= SA_GUARD_MIN_SAMPLE || k >= g->want || g->checks[slot] >= SA_GUARD_MAX_SMALL_CHECKS) {
        g->verified |= kind;
        g->pending &= ~kind;
        g->window[slot] = g->recheck_left[slot] > 0 ? SA_GUARD_RECHECK_CALLS : 0;
    }
    if (!g->pending) guard_drop_shadows(g);
}

/* a verified kind, a host-memory call: does this batch show a status code or a failure regime the verified sample
   never had?  Then the kind is pending again (once) and the caller runs the check on THIS batch. */
static bool guard_recheck_due(sa_solver *s, uint32_t kind, int32_t B, const int32_t *status, const int64_t *stats)
{
    Guard *g = s->guard;
    if (!g || g->using_safe || !(g->verified & kind)) return false;
    const int slot = kind_slot(kind);
    if (g->recheck_left[slot] <= 0 || g->window[slot] <= 0) return false;
    g->window[slot]--;
    const GuardSeen &seen = g->seen[slot];
    if (!seen.valid || !status) return false;
    bool due = false;
    for (int32_t i = 0; i < B && !due; i++) {
        if (std::find(seen.codes.begin(), seen.codes.end(), status[i]) == seen.codes.end()) due = true;
        if (stats && !due) {
            const int64_t *c = stats + (size_t)i * SA_N_STATS;
            due = (c[SA_ST_NETF] > 0 && seen.max_netf == 0) || (c[SA_ST_NCFN] > 0 && seen.max_ncfn == 0) ||
                  (c[SA_ST_NST] > 2 * seen.max_nst && seen.max_nst > 0);
        }
    }
    if (!due) return false;
    g->recheck_left[slot]--;
    g->window[slot] = 0;
    g->verified &= ~kind;
    g->pending |= kind;
    g->checks[slot] = 0;
    return true;
}

/* after the main forward launch of a batch of `mode`: its sample through both shadows */
static int guard_forward(sa_solver *s, int mode, int32_t B, const double *d_y0, const double *d_ps, const double *d_pr,
                         int32_t rem_stride, double t0, const double *d_tv, int32_t n_t, const int32_t *d_status,
                         const int64_t *d_stats)
{
    Guard *g = s->guard;
    if (mode != SA_MODE_PLAIN) {
        /* a forward pass nobody followed with a backward pass cannot finish the adjoint check: after
           SA_GUARD_MAX_FWD_ONLY of them stop paying for the shadows until a backward call has been seen */
        if (g->adj_k > 0) g->fwd_only++;
        g->adj_k = 0;
        if (g->wait_backward) return SA_OK;
        if (g->fwd_only >= SA_GUARD_MAX_FWD_ONLY) { g->wait_backward = true; guard_drop_shadows(g); return SA_OK; }
    }
    int rc;
    if ((rc = guard_shadows(s))) return rc;
    if ((rc = guard_sample(s, B, d_status, d_stats))) return rc;
    const int32_t k = (int32_t)g->idx.size();
    const size_t nn = (size_t)(s->n > 0 ? s->n : 1);
    const void *in_y0, *in_ps, *in_pr;
    if ((rc = guard_gather(s, g->in[0], d_y0, sizeof(double) * (size_t)s->n, true, &in_y0))) return rc;
    if ((rc = guard_gather(s, g->in[1], d_ps, sizeof(double) * (size_t)s->p, true, &in_ps))) return rc;
    if ((rc = guard_gather(s, g->in[2], d_pr, sizeof(double) * (size_t)s->r, rem_stride != 0, &in_pr))) return rc;
    HIP_TRY(hipStreamSynchronize(s->stream));
    const GuardBuf bufs[3] = {{"y_out", sizeof(double) * (size_t)n_t * nn, sizeof(double) * (size_t)n_t * (size_t)s->n},
                              {"status", sizeof(int32_t), sizeof(int32_t)},
                              {"the counters", sizeof(int64_t) * SA_N_STATS, sizeof(int64_t) * 15}};
    sa_solver *sh[2] = {g->fast, g->safe};
    for (int w = 0; w < 2; w++) {
        for (int b = 0; b < 3; b++)
            if ((rc = g->out[w][b].ensure(bufs[b].row_bytes * (size_t)k))) return rc;
        if ((rc = forward_common(sh[w], mode, SA_MEM_DEVICE, k, (const double *)in_y0, (const double *)in_ps,
                                 (const double *)in_pr, rem_stride, t0, d_tv, n_t,
                                 (double *)g->out[w][0].p, (int32_t *)g->out[w][1].p, (int64_t *)g->out[w][2].p)))
            return rc;
    }
    bool same = true;
    if ((rc = guard_compare(s, mode == SA_MODE_PLAIN ? "forward solve" : "adjoint solve, forward pass", k, bufs, 3, &same)))
        return rc;
    if (same && (rc = guard_remember(s, mode == SA_MODE_PLAIN ? SA_GUARD_PLAIN : SA_GUARD_ADJOINT, k, 1, 2))) return rc;
    if (mode == SA_MODE_PLAIN) guard_book(s, SA_GUARD_PLAIN, k, same);
    else if (!same) guard_book(s, SA_GUARD_ADJOINT, k, false);
    else g->adj_k = k;                                   /* the backward call finishes the check */
    return SA_OK;
}

static int guard_sens(sa_solver *s, int ism, const double *scaling, int32_t B, const double *d_y0, const double *d_ps,
                      const double *d_pr, int32_t rem_stride, const double *d_s0, double t0, const double *d_tv,
                      int32_t n_t, const int32_t *d_status, const int64_t *d_stats)
{
    Guard *g = s->guard;
    int rc;
    if ((rc = guard_shadows(s))) return rc;
    if ((rc = guard_sample(s, B, d_status, d_stats))) return rc;
    const int32_t k = (int32_t)g->idx.size();
    const size_t nn = (size_t)(s->n > 0 ? s->n : 1), np_n = (size_t)s->p * (size_t)s->n;
    const void *in_y0, *in_ps, *in_pr, *in_s0;
    if ((rc = guard_gather(s, g->in[0], d_y0, sizeof(double) * (size_t)s->n, true, &in_y0))) return rc;
    if ((rc = guard_gather(s, g->in[1], d_ps, sizeof(double) * (size_t)s->p, true, &in_ps))) return rc;
    if ((rc = guard_gather(s, g->in[2], d_pr, sizeof(double) * (size_t)s->r, rem_stride != 0, &in_pr))) return rc;
    if ((rc = guard_gather(s, g->in[3], d_s0, sizeof(double) * np_n, true, &in_s0))) return rc;
    HIP_TRY(hipStreamSynchronize(s->stream));
    const GuardBuf bufs[4] = {{"y_out", sizeof(double) * (size_t)n_t * nn, sizeof(double) * (size_t)n_t * (size_t)s->n},
                              {"sens_out", sizeof(double) * (size_t)n_t * (np_n ? np_n : 1), sizeof(double) * (size_t)n_t * np_n},
                              {"status", sizeof(int32_t), sizeof(int32_t)},
                              {"the counters", sizeof(int64_t) * SA_N_STATS, sizeof(int64_t) * 15}};
    sa_solver *sh[2] = {g->fast, g->safe};
    for (int w = 0; w < 2; w++) {
        for (int b = 0; b < 4; b++)
            if ((rc = g->out[w][b].ensure(bufs[b].row_bytes * (size_t)k))) return rc;
        if ((rc = sa_solve_sens_batch(sh[w], SA_MEM_DEVICE, ism, scaling, k, (const double *)in_y0, (const double *)in_ps,
                                      (const double *)in_pr, rem_stride, (const double *)in_s0, t0, d_tv,
                                      n_t, (double *)g->out[w][0].p, (double *)g->out[w][1].p, (int32_t *)g->out[w][2].p,
                                      (int64_t *)g->out[w][3].p)))
            return rc;
    }
    bool same = true;
    if ((rc = guard_compare(s, "forward-sensitivity solve", k, bufs, 4, &same))) return rc;
    if (same && (rc = guard_remember(s, SA_GUARD_SENS, k, 2, 3))) return rc;
    guard_book(s, SA_GUARD_SENS, k, same);
    return SA_OK;
}

static int guard_backward(sa_solver *s, int32_t B, const double *d_ps, const double *d_pr, int32_t rem_stride,
                          double t0, double tend, const double *d_tv, int32_t n_t, const double *d_g,
                          int64_t grads_stride)
{
    Guard *g = s->guard;
    if (g->wait_backward) { g->wait_backward = false; g->fwd_only = 0; }       /* the next forward call checks again */
    const int32_t k = g->adj_k;
    if (k <= 0 || k > B || !g->fast || !g->safe || (int32_t)g->idx.size() != k) return SA_OK;    /* no forward check in flight for this batch */
    int rc;
    const size_t nn = (size_t)(s->n > 0 ? s->n : 1), pp = (size_t)(s->p > 0 ? s->p : 1);
    /* the sample's parameters were gathered by the forward check (g->in[1], g->in[2]); its cotangents now */
    const void *in_ps = g->prefix ? (const void *)d_ps : g->in[1].p;
    const void *in_pr = (g->prefix || rem_stride == 0) ? (const void *)d_pr : g->in[2].p;
    const void *in_g;
    if ((rc = guard_gather(s, g->in[4], d_g, sizeof(double) * (size_t)n_t * (size_t)s->n, grads_stride != 0, &in_g))) return rc;
    HIP_TRY(hipStreamSynchronize(s->stream));
    const GuardBuf bufs[4] = {{"grad_out", sizeof(double) * pp, sizeof(double) * (size_t)s->p},
                              {"lamda_out", sizeof(double) * nn, sizeof(double) * (size_t)s->n},
                              {"status", sizeof(int32_t), sizeof(int32_t)},
                              {"the counters", sizeof(int64_t) * SA_N_STATS, sizeof(int64_t) * 15}};
    sa_solver *sh[2] = {g->fast, g->safe};
    for (int w = 0; w < 2; w++) {
        for (int b = 0; b < 4; b++)
            if ((rc = g->out[w][b].ensure(bufs[b].row_bytes * (size_t)k))) return rc;
        if ((rc = sa_solve_backward_batch_all(sh[w], SA_MEM_DEVICE, k, (const double *)in_ps, (const double *)in_pr,
                                              rem_stride, t0, tend, d_tv, n_t, (const double *)in_g,
                                              grads_stride, (double *)g->out[w][0].p, (double *)g->out[w][1].p, nullptr,
                                              nullptr, (int32_t *)g->out[w][2].p, (int64_t *)g->out[w][3].p)))
            return rc;
    }
    bool same = true;
    if ((rc = guard_compare(s, "adjoint solve, backward pass", k, bufs, 4, &same))) return rc;
    g->adj_k = 0;
    g->fwd_only = 0;
    guard_book(s, SA_GUARD_ADJOINT, k, same);
    if (!same) {
        (void)guard_switched(s);
        /* the forward pass of THIS batch ran on the default build (whose sample agreed with the conservative
           build's): integrate it again with the code object the handle has just switched to, then go backward.
           (The y_out / status the caller received from that forward call came from the default build and are NOT
           recomputed -- sunode_amd's RuntimeWarning says so.) */
        const size_t nB = (size_t)s->fwd_B;
        if ((rc = s->t_yout.ensure(sizeof(double) * nB * (size_t)s->fwd_n_t * nn))) return rc;
        if ((rc = s->t_status.ensure(sizeof(int32_t) * (size_t)round64(s->fwd_B)))) return rc;
        if ((rc = s->t_stats.ensure(sizeof(int64_t) * (size_t)round64(s->fwd_B) * SA_N_STATS))) return rc;
        if ((rc = forward_common(s, SA_MODE_ADJ_FWD, SA_MEM_DEVICE, s->fwd_B, (const double *)s->keep_y0.p, d_ps, d_pr,
                                 rem_stride, s->fwd_t0, (const double *)s->keep_tvals.p, s->fwd_n_t,
                                 (double *)s->t_yout.p, (int32_t *)s->t_status.p, (int64_t *)s->t_stats.p)))
            return rc;
        if ((rc = resolve_forward(s))) return rc;
    }
    return SA_OK;
}

extern "C" int sa_solver_attach_guard(sa_solver *s, const char *safe_path, int32_t n_sample, uint32_t verified_kinds)
{
    if (!s || !safe_path) return fail(SA_ERR_ARG, "null argument");
    if (n_sample < 0) return fail(SA_ERR_ARG, "n_sample must be >= 0");
    DeviceScope scope(s->device);
    guard_free(s);
    {   /* fail HERE if the conservative code object is missing, not in the first solve (the module itself is loaded
           when the first check is due) */
        FILE *fh = fopen(safe_path, "rb");
        if (!fh) return fail(SA_ERR_MODULE, "guard: cannot read the conservative code object %s", safe_path);
        fclose(fh);
    }
    Guard *g = new Guard();
    g->safe_path = safe_path;
    g->want = n_sample > 0 ? n_sample : 64;
    const uint32_t all = SA_GUARD_PLAIN | SA_GUARD_ADJOINT | (s->k_sens ? SA_GUARD_SENS : 0u);
    g->verified = verified_kinds & all;
    g->pending = all & ~g->verified;
    s->guard = g;
    return SA_OK;
}

extern "C" int sa_guard_state(sa_solver *s, uint32_t *pending, uint32_t *verified, uint32_t *differs,
                              int32_t *using_safe, int32_t *n_sample, const char **detail)
{
    if (!s) return fail(SA_ERR_ARG, "null solver");
    static const char *none = "";
    Guard *g = s->guard;
    uint32_t open = 0;
    if (g && !g->using_safe)
        for (int i = 0; i < 3; i++) if (g->recheck_left[i] > 0 && g->window[i] > 0) open = SA_GUARD_RECHECK_OPEN;
    if (pending) *pending = g ? (g->pending | open) : 0;
    if (verified) *verified = g ? g->verified : 0;
    if (differs) *differs = g ? g->differs : 0;
    if (using_safe) *using_safe = g && g->using_safe ? 1 : 0;
    if (n_sample) for (int i = 0; i < 3; i++) n_sample[i] = g ? g->best[i] : 0;
    if (detail) *detail = g ? g->detail.c_str() : none;
    return SA_OK;
}

extern "C" int sa_eval_callbacks(sa_solver *s, int mem, int32_t npts, const double *t, const double *y,
                                 const double *lam, const double *ps, const double *pr, double *rhs, double *jac,
                                 double *adj, double *quad, double *adjjac, int32_t *codes)
{
    if (!s) return fail(SA_ERR_ARG, "null solver");
    HIP_TRY(hipSetDevice(s->device));
    if (npts <= 0) return SA_OK;
    const size_t N = (size_t)npts, n = s->n, p = s->p, r = s->r;
    sa_eval_args a;
    memset(&a, 0, sizeof a);
    a.npts = npts;
    int rc;
    if (mem == SA_MEM_HOST) {
        const void *q;
        if ((rc = stage_in(s, s->s_misc[0], t, sizeof(double) * N, &q))) return rc; a.t = (const double *)q;
        if ((rc = stage_in(s, s->s_misc[1], y, sizeof(double) * N * n, &q))) return rc; a.y = (const double *)q;
        if ((rc = stage_in(s, s->s_misc[2], lam, sizeof(double) * N * n, &q))) return rc; a.lam = (const double *)q;
        if ((rc = stage_in(s, s->s_misc[3], ps, sizeof(double) * N * p, &q))) return rc; a.ps = (const double *)q;
        if ((rc = stage_in(s, s->s_misc[4], pr, sizeof(double) * N * r, &q))) return rc; a.pr = (const double *)q;
        size_t sz[6] = {sizeof(double) * N * n, sizeof(double) * N * n * n, sizeof(double) * N * n,
                        sizeof(double) * N * p, sizeof(double) * N * n * n, sizeof(int32_t) * N * 5};
        for (int i = 0; i < 6; i++) if ((rc = s->s_misc[5 + i].ensure(sz[i] ? sz[i] : 8))) return rc;
        a.rhs = (double *)s->s_misc[5].p; a.jac = (double *)s->s_misc[6].p; a.adj = (double *)s->s_misc[7].p;
        a.quad = (double *)s->s_misc[8].p; a.adjjac = (double *)s->s_misc[9].p; a.codes = (int32_t *)s->s_misc[10].p;
        if ((rc = launch(s, s->k_eval, npts, &a, sizeof a))) return rc;
        void *host[6] = {rhs, jac, adj, quad, adjjac, codes};
        for (int i = 0; i < 6; i++)
            if (sz[i] && host[i]) HIP_TRY(hipMemcpyAsync(host[i], s->s_misc[5 + i].p, sz[i], hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
        return SA_OK;
    }
    a.t = t; a.y = y; a.lam = lam; a.ps = ps; a.pr = pr;
    a.rhs = rhs; a.jac = jac; a.adj = adj; a.quad = quad; a.adjjac = adjjac; a.codes = codes;
    return launch(s, s->k_eval, npts, &a, sizeof a);
}

extern "C" int sa_math_probe(sa_solver *s, int32_t n, const double *x, const double *y, double *pow_out,
                             double *sqrt_out, double *div_out)
{
    if (!s) return fail(SA_ERR_ARG, "null solver");
    HIP_TRY(hipSetDevice(s->device));
    if (n <= 0) return SA_OK;
    const size_t bytes = sizeof(double) * (size_t)n;
    sa_math_args a;
    memset(&a, 0, sizeof a);
    a.n = n;
    const void *q;
    int rc;
    if ((rc = stage_in(s, s->s_misc[0], x, bytes, &q))) return rc; a.x = (const double *)q;
    if ((rc = stage_in(s, s->s_misc[1], y, bytes, &q))) return rc; a.y = (const double *)q;
    for (int i = 0; i < 3; i++) if ((rc = s->s_misc[5 + i].ensure(bytes))) return rc;
    a.pow_out = (double *)s->s_misc[5].p; a.sqrt_out = (double *)s->s_misc[6].p; a.div_out = (double *)s->s_misc[7].p;
    if ((rc = launch(s, s->k_math, n, &a, sizeof a))) return rc;
    HIP_TRY(hipMemcpyAsync(pow_out, a.pow_out, bytes, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipMemcpyAsync(sqrt_out, a.sqrt_out, bytes, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipMemcpyAsync(div_out, a.div_out, bytes, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return SA_OK;
}
