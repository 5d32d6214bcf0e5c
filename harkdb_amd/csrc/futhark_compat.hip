// futhark_compat.hip -- the generated-Futhark-C-API names (include/futhark_compat.h) over the hark_* ABI.
// What FutharkContext.py:41,65-66,70-71 reaches through futhark_ffi.Futhark(_main) lands here unchanged.
#include "hark_internal.h"
#include "../../include/futhark_compat.h"

struct futhark_context_config {
    int device = 0;
    int debugging = 0, profiling = 0, logging = 0;
};

struct futhark_context {
    hark_context *h = nullptr;
    std::string err;            // failure of the veneer itself (no device, bad argument)
    bool failed = false;
};

// A 2-d array is a device-resident table (made by futhark_new_*_2d) or the result of an entry.
struct fut_arr2d {
    hark_table *t = nullptr;
    hark_result *r = nullptr;
    int64_t shape[2] = {0, 0};
};
struct futhark_i32_2d { fut_arr2d a; };
struct futhark_u32_2d { fut_arr2d a; };
// The 1-d arrays of main.fut are column-index lists: the hark entries take them as host int32 arrays.
struct futhark_i32_1d {
    std::vector<int32_t> v;
    int64_t shape[1] = {0};
};

namespace {

int veneer_fail(futhark_context *ctx, const char *msg)
{
    if (ctx) { ctx->err = msg; ctx->failed = true; }
    return HARK_EARG;
}

// A borrowed hark_table over whichever object backs the array (columns not owned, nothing copied).
void view_of(const fut_arr2d &a, hark_table *view)
{
    view->n = a.shape[0];
    view->m = a.shape[1];
    view->cols.clear();
    const std::vector<hark_column> &src = a.t ? a.t->cols : a.r->cols;
    for (const hark_column &c : src) { hark_column b; b.data = c.data; b.dtype = c.dtype; b.owned = false; view->cols.push_back(b); }
}

fut_arr2d *new_2d(futhark_context *ctx, fut_arr2d *a, const void *data, int dtype, int64_t d0, int64_t d1)
{
    if (!ctx || !ctx->h) return nullptr;
    if (hark_table_new_2d(ctx->h, &a->t, data, dtype, d0, d1, /*row_stride*/ d1, /*col_stride*/ 1) != HARK_OK) { ctx->failed = true; return nullptr; }
    a->shape[0] = d0; a->shape[1] = d1;
    return a;
}

int free_2d(futhark_context *ctx, fut_arr2d *a)
{
    if (!ctx || !ctx->h || !a) return HARK_EARG;
    if (a->t) hark_table_free(ctx->h, a->t);
    if (a->r) hark_result_free(ctx->h, a->r);
    a->t = nullptr; a->r = nullptr;
    return HARK_OK;
}

int values_2d(futhark_context *ctx, fut_arr2d *a, void *data, int dtype)
{
    if (!ctx || !ctx->h || !a || (!a->t && !a->r)) return veneer_fail(ctx, "futhark_values: null array");
    if (a->r) return hark_result_values_2d(ctx->h, a->r, data, dtype);
    hark_result tmp;                                         // borrow the table's columns as a result
    tmp.n = a->t->n;
    for (const hark_column &c : a->t->cols) { hark_column b; b.data = c.data; b.dtype = c.dtype; b.owned = false; tmp.cols.push_back(b); }
    return hark_result_values_2d(ctx->h, &tmp, data, dtype);
}

void adopt(fut_arr2d *a, hark_result *r)
{
    a->r = r;
    int64_t n = 0, m = 0;
    hark_result_shape(r, &n, &m);
    a->shape[0] = n; a->shape[1] = m;
}

} // namespace

extern "C" {

struct futhark_context_config *futhark_context_config_new(void) { return new futhark_context_config(); }
void futhark_context_config_free(struct futhark_context_config *cfg) { delete cfg; }
void futhark_context_config_set_debugging(struct futhark_context_config *cfg, int flag) { if (cfg) cfg->debugging = flag; }
void futhark_context_config_set_profiling(struct futhark_context_config *cfg, int flag) { if (cfg) cfg->profiling = flag; }
void futhark_context_config_set_logging(struct futhark_context_config *cfg, int flag) { if (cfg) cfg->logging = flag; }
void futhark_context_config_set_device(struct futhark_context_config *cfg, const char *s)
{
    if (!cfg || !s) return;
    if (*s == '#') s++;
    cfg->device = atoi(s);
}

struct futhark_context *futhark_context_new(struct futhark_context_config *cfg)
{
    futhark_context *ctx = new futhark_context();
    const int rc = hark_context_new(&ctx->h, cfg ? cfg->device : 0);
    if (rc != HARK_OK) {                                     // like Futhark: the context exists and carries the error
        ctx->h = nullptr;
        ctx->failed = true;
        ctx->err = "futhark_context_new: no usable HIP device (libhark.so has no CPU fallback)";
    }
    return ctx;
}

void futhark_context_free(struct futhark_context *ctx)
{
    if (!ctx) return;
    if (ctx->h) hark_context_free(ctx->h);
    delete ctx;
}

int futhark_context_sync(struct futhark_context *ctx)
{
    if (!ctx || !ctx->h) return HARK_EARG;
    return hark_context_sync(ctx->h);
}

char *futhark_context_get_error(struct futhark_context *ctx)
{
    if (!ctx) return nullptr;
    std::string msg;
    if (!ctx->err.empty()) { msg = ctx->err; ctx->err.clear(); }
    else if (ctx->h) { const char *e = hark_context_get_error(ctx->h); if (e && *e) msg = e; }
    if (msg.empty()) return nullptr;
    char *out = static_cast<char *>(malloc(msg.size() + 1));
    if (out) memcpy(out, msg.c_str(), msg.size() + 1);
    return out;
}

char *futhark_context_report(struct futhark_context *ctx)
{
    (void)ctx;
    char *out = static_cast<char *>(malloc(1));
    if (out) out[0] = 0;
    return out;
}

int futhark_context_clear_caches(struct futhark_context *ctx)
{
    if (!ctx || !ctx->h) return HARK_EARG;
    return hark_context_trim(ctx->h);
}

void futhark_context_pause_profiling(struct futhark_context *ctx) { (void)ctx; }
void futhark_context_unpause_profiling(struct futhark_context *ctx) { (void)ctx; }

// ---- []i32 ---------------------------------------------------------------------------------------
struct futhark_i32_1d *futhark_new_i32_1d(struct futhark_context *ctx, const int32_t *data, int64_t dim0)
{
    if (!ctx || dim0 < 0 || (dim0 && !data)) return nullptr;
    futhark_i32_1d *a = new futhark_i32_1d();
    a->v.assign(data, data + dim0);
    a->shape[0] = dim0;
    return a;
}
int futhark_free_i32_1d(struct futhark_context *ctx, struct futhark_i32_1d *arr) { (void)ctx; delete arr; return 0; }
int futhark_values_i32_1d(struct futhark_context *ctx, struct futhark_i32_1d *arr, int32_t *data)
{
    if (!arr || (arr->shape[0] && !data)) return veneer_fail(ctx, "futhark_values_i32_1d: null argument");
    if (arr->shape[0]) memcpy(data, arr->v.data(), (size_t)arr->shape[0] * 4);
    return 0;
}
const int64_t *futhark_shape_i32_1d(struct futhark_context *ctx, struct futhark_i32_1d *arr) { (void)ctx; return arr ? arr->shape : nullptr; }

// ---- [][]i32 / [][]u32 ---------------------------------------------------------------------------
struct futhark_i32_2d *futhark_new_i32_2d(struct futhark_context *ctx, const int32_t *data, int64_t dim0, int64_t dim1)
{
    futhark_i32_2d *a = new futhark_i32_2d();
    if (!new_2d(ctx, &a->a, data, HARK_I32, dim0, dim1)) { delete a; return nullptr; }
    return a;
}
int futhark_free_i32_2d(struct futhark_context *ctx, struct futhark_i32_2d *arr) { if (!arr) return 0; const int rc = free_2d(ctx, &arr->a); delete arr; return rc; }
int futhark_values_i32_2d(struct futhark_context *ctx, struct futhark_i32_2d *arr, int32_t *data) { return values_2d(ctx, arr ? &arr->a : nullptr, data, HARK_I32); }
const int64_t *futhark_shape_i32_2d(struct futhark_context *ctx, struct futhark_i32_2d *arr) { (void)ctx; return arr ? arr->a.shape : nullptr; }

struct futhark_u32_2d *futhark_new_u32_2d(struct futhark_context *ctx, const uint32_t *data, int64_t dim0, int64_t dim1)
{
    futhark_u32_2d *a = new futhark_u32_2d();
    if (!new_2d(ctx, &a->a, data, HARK_U32, dim0, dim1)) { delete a; return nullptr; }
    return a;
}
int futhark_free_u32_2d(struct futhark_context *ctx, struct futhark_u32_2d *arr) { if (!arr) return 0; const int rc = free_2d(ctx, &arr->a); delete arr; return rc; }
int futhark_values_u32_2d(struct futhark_context *ctx, struct futhark_u32_2d *arr, uint32_t *data) { return values_2d(ctx, arr ? &arr->a : nullptr, data, HARK_U32); }
const int64_t *futhark_shape_u32_2d(struct futhark_context *ctx, struct futhark_u32_2d *arr) { (void)ctx; return arr ? arr->a.shape : nullptr; }

// ---- entries --------------------------------------------------------------------------------------
int futhark_entry_query_sel(struct futhark_context *ctx, struct futhark_i32_2d **out0,
                            const struct futhark_i32_2d *in0, const struct futhark_i32_1d *in1)
{
    if (!ctx || !ctx->h || !out0 || !in0 || !in1) return veneer_fail(ctx, "futhark_entry_query_sel: null argument");
    *out0 = nullptr;
    hark_table view;
    view_of(in0->a, &view);
    hark_result *r = nullptr;
    const int rc = hark_entry_query_sel(ctx->h, &r, &view, in1->v.data(), in1->shape[0]);
    if (rc != HARK_OK) return rc;
    futhark_i32_2d *o = new futhark_i32_2d();
    adopt(&o->a, r);
    *out0 = o;
    return 0;
}

int futhark_entry_query_groupby(struct futhark_context *ctx, struct futhark_u32_2d **out0,
                                const struct futhark_u32_2d *in0, const int32_t in1,
                                const struct futhark_i32_1d *in2, const struct futhark_i32_1d *in3)
{
    if (!ctx || !ctx->h || !out0 || !in0 || !in2 || !in3) return veneer_fail(ctx, "futhark_entry_query_groupby: null argument");
    *out0 = nullptr;
    hark_table view;
    view_of(in0->a, &view);
    hark_result *r = nullptr;
    const int rc = hark_entry_query_groupby(ctx->h, &r, &view, in1, in2->v.data(), in2->shape[0], in3->v.data(), in3->shape[0]);
    if (rc != HARK_OK) return rc;
    futhark_u32_2d *o = new futhark_u32_2d();
    adopt(&o->a, r);
    *out0 = o;
    return 0;
}

int futhark_entry_join(struct futhark_context *ctx, struct futhark_u32_2d **out0,
                       const struct futhark_u32_2d *in0, const struct futhark_u32_2d *in1,
                       const int32_t in2, const int32_t in3,
                       const struct futhark_i32_1d *in4, const struct futhark_i32_1d *in5)
{
    if (!ctx || !ctx->h || !out0 || !in0 || !in1 || !in4 || !in5) return veneer_fail(ctx, "futhark_entry_join: null argument");
    *out0 = nullptr;
    hark_table v0, v1;
    view_of(in0->a, &v0);
    view_of(in1->a, &v1);
    hark_result *r = nullptr;
    const int rc = hark_entry_join(ctx->h, &r, &v0, &v1, in2, in3, in4->v.data(), in4->shape[0], in5->v.data(), in5->shape[0]);
    if (rc != HARK_OK) return rc;
    futhark_u32_2d *o = new futhark_u32_2d();
    adopt(&o->a, r);
    *out0 = o;
    return 0;
}

} // extern "C"
