// Context, error reporting and the Field part of the C ABI (include/mgcr.h).
#include <cstdarg>

#include "internal.h"

namespace mgcr {

static thread_local std::string g_err;

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

int hip_fail(hipError_t e, const char *what, const char *file, int line) {
    set_error("HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    return MGCR_ERR_HIP;
}

Context &ctx() {
    static Context c;
    return c;
}

int require_ctx() {
    if (!ctx().ready) {
        set_error("mgcr_init() has not been called or no HIP device is usable (there is no CPU fallback)");
        return MGCR_ERR_NO_DEVICE;
    }
    return MGCR_OK;
}

}  // namespace mgcr

using namespace mgcr;

extern "C" {

const char *mgcr_last_error(void) { return g_err.c_str(); }
const char *mgcr_version(void) { return "mgcr-hip 0.1 (gfx950)"; }

int mgcr_init(int device) {
    Context &c = ctx();
    std::lock_guard<std::recursive_mutex> lk(c.mtx);
    if (c.ready) {
        if (c.device == device) return MGCR_OK;
        set_error("mgcr_init: already initialised on device %d", c.device);
        return MGCR_ERR_INVALID;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        set_error("mgcr_init: no HIP device visible (%s); this library has no CPU fallback",
                  e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return MGCR_ERR_NO_DEVICE;
    }
    MGCR_CHECK(device >= 0 && device < ndev, MGCR_ERR_INVALID, "mgcr_init: device %d out of range [0,%d)", device, ndev);
    MGCR_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    MGCR_HIP(hipGetDeviceProperties(&prop, device));
    c.n_cu = prop.multiProcessorCount;
    MGCR_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    MGCR_HIP(hipEventCreate(&c.ev0));
    MGCR_HIP(hipEventCreate(&c.ev1));
    MGCR_HIP(hipHostMalloc((void **)&c.h_mail, 64 * sizeof(double), hipHostMallocDefault));
    c.device = device;
    c.ready = true;
    return MGCR_OK;
}

int mgcr_finalize(void) {
    Context &c = ctx();
    std::lock_guard<std::recursive_mutex> lk(c.mtx);
    if (!c.ready) return MGCR_OK;
    hipStreamSynchronize(c.stream);
    resident_shutdown();
    hipEventDestroy(c.ev0);
    hipEventDestroy(c.ev1);
    hipHostFree(c.h_mail);
    hipStreamDestroy(c.stream);
    c.ready = false;
    c.device = -1;
    return MGCR_OK;
}

int mgcr_device_info(char *name, int name_cap, int *n_cu, int64_t *mem_bytes) {
    MGCR_TRY(require_ctx());
    hipDeviceProp_t prop;
    MGCR_HIP(hipGetDeviceProperties(&prop, ctx().device));
    if (name && name_cap > 0) snprintf(name, (size_t)name_cap, "%s (%s)", prop.name, prop.gcnArchName);
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (mem_bytes) *mem_bytes = (int64_t)prop.totalGlobalMem;
    return MGCR_OK;
}

int mgcr_synchronize(void) {
    MGCR_TRY(require_ctx());
    MGCR_HIP(hipStreamSynchronize(ctx().stream));
    MGCR_TRY(comm_check_all());   // a peer-write wait that timed out in any kernel enqueued so far
    return resident_check();      // a one-launch solve that was not co-resident and gave up
}

int mgcr_timer_start(void) {
    MGCR_TRY(require_ctx());
    MGCR_HIP(hipEventRecord(ctx().ev0, ctx().stream));
    return MGCR_OK;
}
int mgcr_timer_stop(double *ms) {
    MGCR_TRY(require_ctx());
    MGCR_HIP(hipEventRecord(ctx().ev1, ctx().stream));
    MGCR_HIP(hipEventSynchronize(ctx().ev1));
    float f = 0.f;
    MGCR_HIP(hipEventElapsedTime(&f, ctx().ev0, ctx().ev1));
    if (ms) *ms = (double)f;
    return MGCR_OK;
}

// ---- Field ------------------------------------------------------------------------------------
#define LOCK() std::lock_guard<std::recursive_mutex> lk__(ctx().mtx)

int mgcr_vec_create(int64_t n, mgcr_vec_t *out) {
    MGCR_TRY(require_ctx());
    MGCR_CHECK(out && n >= 0, MGCR_ERR_INVALID, "mgcr_vec_create: bad arguments");
    LOCK();
    mgcr_vec_s *v = new mgcr_vec_s();
    v->n = n;
    if (n > 0) {
        hipError_t e = hipMalloc((void **)&v->d, sizeof(cplx) * (size_t)n);
        if (e != hipSuccess) {
            delete v;
            set_error("mgcr_vec_create: hipMalloc of %lld complex failed: %s", (long long)n, hipGetErrorString(e));
            return MGCR_ERR_ALLOC;
        }
    }
    *out = v;
    return MGCR_OK;
}

int mgcr_vec_destroy(mgcr_vec_t v) {
    if (!v) return MGCR_OK;
    LOCK();
    if (ctx().ready) hipStreamSynchronize(ctx().stream);
    if (v->owns && v->d) hipFree(v->d);
    delete v;
    return MGCR_OK;
}

int64_t mgcr_vec_size(mgcr_vec_t v) { return v ? v->n : -1; }

int mgcr_vec_upload(mgcr_vec_t v, const double *host_ri) {
    MGCR_TRY(require_ctx());
    MGCR_CHECK(v && host_ri, MGCR_ERR_INVALID, "mgcr_vec_upload: null argument");
    LOCK();
    MGCR_HIP(hipMemcpyAsync(v->w(), host_ri, sizeof(cplx) * (size_t)v->n, hipMemcpyHostToDevice, ctx().stream));
    MGCR_HIP(hipStreamSynchronize(ctx().stream));
    return MGCR_OK;
}

int mgcr_vec_download(mgcr_vec_t v, double *host_ri) {
    MGCR_TRY(require_ctx());
    MGCR_CHECK(v && host_ri, MGCR_ERR_INVALID, "mgcr_vec_download: null argument");
    LOCK();
    MGCR_HIP(hipMemcpyAsync(host_ri, v->d, sizeof(cplx) * (size_t)v->n, hipMemcpyDeviceToHost, ctx().stream));
    MGCR_HIP(hipStreamSynchronize(ctx().stream));
    MGCR_TRY(comm_check_all());
    return resident_check();
}

int mgcr_vec_copy(mgcr_vec_t dst, mgcr_vec_t src) {
    MGCR_TRY(require_ctx());
    MGCR_CHECK(dst && src, MGCR_ERR_INVALID, "mgcr_vec_copy: null argument");
    // Field::operator= on a size mismatch prints "Dimension mismatch." and exits (src/Fields.h:279-283)
    MGCR_CHECK(dst->n == src->n, MGCR_ERR_INVALID, "Dimension mismatch. (%lld vs %lld)", (long long)dst->n, (long long)src->n);
    LOCK();
    return k_copy(dst->w(), src->d, dst->n);
}

int mgcr_vec_zero(mgcr_vec_t v) {
    MGCR_TRY(require_ctx());
    MGCR_CHECK(v, MGCR_ERR_INVALID, "mgcr_vec_zero: null argument");
    LOCK();
    MGCR_TRY(k_zero(v->w(), v->n));
    v->zero_known = true;
    return MGCR_OK;
}

int mgcr_vec_set_constant(mgcr_vec_t v, const double c_ri[2]) {
    MGCR_TRY(require_ctx());
    MGCR_CHECK(v && c_ri, MGCR_ERR_INVALID, "mgcr_vec_set_constant: null argument");
    LOCK();
    return k_set_constant(v->w(), make_double2(c_ri[0], c_ri[1]), v->n);
}

int mgcr_vec_fill_rhs(mgcr_vec_t v, uint64_t seed, int64_t global_offset) {
    MGCR_TRY(require_ctx());
    MGCR_CHECK(v, MGCR_ERR_INVALID, "mgcr_vec_fill_rhs: null argument");
    LOCK();
    return k_fill_rhs(v->w(), v->n, seed, global_offset);
}

static int dot_to_host(const cplx *a, const cplx *b, int64_t n, double out[2]) {
    Context &c = ctx();
    static double *parts = nullptr, *dres = nullptr;
    if (!parts) {
        MGCR_HIP(hipMalloc((void **)&parts, sizeof(double) * 2 * RED_MAX_BLOCKS));
        MGCR_HIP(hipMalloc((void **)&dres, sizeof(double) * 2));
    }
    int nblk = 0;
    MGCR_TRY(k_dot_partials(a, b, n, parts, &nblk));
    MGCR_TRY(k_fold(parts, nblk, 2, dres));
    MGCR_HIP(hipMemcpyAsync(c.h_mail, dres, 2 * sizeof(double), hipMemcpyDeviceToHost, c.stream));
    MGCR_HIP(hipStreamSynchronize(c.stream));
    out[0] = c.h_mail[0];
    out[1] = c.h_mail[1];
    MGCR_TRY(comm_check_all());
    return resident_check();
}

int mgcr_dot(mgcr_vec_t a, mgcr_vec_t b, double out_ri[2]) {
    MGCR_TRY(require_ctx());
    MGCR_CHECK(a && b && out_ri, MGCR_ERR_INVALID, "mgcr_dot: null argument");
    MGCR_CHECK(a->n == b->n, MGCR_ERR_INVALID, "Lengths of two fields do not match!");
    LOCK();
    return dot_to_host(a->d, b->d, a->n, out_ri);
}

int mgcr_norm2(mgcr_vec_t a, double *out) {
    MGCR_TRY(require_ctx());
    MGCR_CHECK(a && out, MGCR_ERR_INVALID, "mgcr_norm2: null argument");
    LOCK();
    double r[2];
    MGCR_TRY(dot_to_host(a->d, a->d, a->n, r));
    *out = r[0];
    return MGCR_OK;
}

int mgcr_add_scaled(mgcr_vec_t out, mgcr_vec_t a, const double alpha_ri[2], mgcr_vec_t b) {
    MGCR_TRY(require_ctx());
    MGCR_CHECK(out && a && b && alpha_ri, MGCR_ERR_INVALID, "mgcr_add_scaled: null argument");
    MGCR_CHECK(out->n == a->n && a->n == b->n, MGCR_ERR_INVALID, "Lengths of two fields do not match!");
    LOCK();
    return k_add_scaled(out->w(), a->d, make_double2(alpha_ri[0], alpha_ri[1]), b->d, a->n);
}

int mgcr_axpy(const double alpha_ri[2], mgcr_vec_t x, mgcr_vec_t y) {
    MGCR_TRY(require_ctx());
    MGCR_CHECK(x && y && alpha_ri, MGCR_ERR_INVALID, "mgcr_axpy: null argument");
    MGCR_CHECK(x->n == y->n, MGCR_ERR_INVALID, "Field dimensions do not match!");
    LOCK();
    return k_add_scaled(y->w(), y->d, make_double2(alpha_ri[0], alpha_ri[1]), x->d, x->n);
}

int mgcr_scale(mgcr_vec_t v, const double alpha_ri[2]) {
    MGCR_TRY(require_ctx());
    MGCR_CHECK(v && alpha_ri, MGCR_ERR_INVALID, "mgcr_scale: null argument");
    LOCK();
    return k_scale(v->w(), make_double2(alpha_ri[0], alpha_ri[1]), v->n);
}

int mgcr_vec_gamma5(mgcr_vec_t in, mgcr_vec_t out, int64_t inner) {
    MGCR_TRY(require_ctx());
    MGCR_CHECK(in && out && in != out && in->n == out->n, MGCR_ERR_INVALID, "mgcr_vec_gamma5: two distinct fields of one size");
    MGCR_CHECK(inner >= 1 && in->n % (4 * inner) == 0, MGCR_ERR_INVALID, "mgcr_vec_gamma5: the spinor dimension must have 4 entries");
    LOCK();
    return k_gamma5(out->w(), in->d, in->n, inner);
}

int mgcr_normalise(mgcr_vec_t v) {
    MGCR_TRY(require_ctx());
    MGCR_CHECK(v, MGCR_ERR_INVALID, "mgcr_normalise: null argument");
    LOCK();
    double r[2];
    MGCR_TRY(dot_to_host(v->d, v->d, v->n, r));
    // field[i] *= 1./norm  (src/Fields.h:237-243)
    return k_scale(v->w(), make_double2(1. / sqrt(r[0]), 0.), v->n);
}

}  // extern "C"
