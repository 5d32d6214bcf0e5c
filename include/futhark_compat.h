/*
 * futhark_compat.h -- the C API `futhark c --library futhark/main.fut -o main` generates
 * (setup.sh:12), under its own names, served by libhark.so.
 *
 * The reference binds exactly these symbols: `build_futhark_ffi main` (setup.sh:13) wraps main.h
 * into the cffi module `_main`, and `futhark_ffi.Futhark(_main)` (FutharkContext.py:31-41) calls
 *     futhark_context_config_new / futhark_context_new            (FutharkContext.py:41)
 *     futhark_new_i32_2d / futhark_new_u32_2d / futhark_new_i32_1d (argument conversion at :65, :70)
 *     futhark_entry_query_sel / futhark_entry_query_groupby        (FutharkContext.py:65, :70;
 *                                                                  futhark/main.fut:7, :9)
 *     futhark_shape_* / futhark_values_* / futhark_free_*          (from_futhark, :66, :71)
 * so a maintainer can point `build_futhark_ffi` (or any C caller of the generated main.h) at this
 * header + libhark.so instead of the generated main.c and keep FutharkContext.py unchanged.
 * Each function is a thin call into the hark_* entry named beside it (include/hark.h).
 *
 * The generated main.h is not in the reference tree (it is a build product and no `futhark`
 * compiler exists in this image): names, argument order and ownership follow the documented
 * Futhark C API (array types futhark_<t>_<r>d; `new` copies from host; `values` copies row-major
 * to host; `shape` returns a pointer owned by the array; entries return 0 on success and write
 * fresh arrays the caller frees; futhark_context_get_error returns a malloc'd string the CALLER
 * frees, or NULL).  Entry argument types come from the Futhark sources:
 *   query_sel     : [][]i32 -> []i32 -> [][]i32                  (main.fut:7, select.fut:23)
 *   query_groupby : [][]u32 -> i32 -> []i32 -> []i32 -> [][]u32  (main.fut:9, groupby.fut:60)
 *   join          : [][]u32 -> [][]u32 -> i32 -> i32 -> []i32 -> []i32 -> [][]u32  (join.fut:52,
 *                   an entry of join.fut itself; main.fut does not import it)
 */
#ifndef FUTHARK_COMPAT_H
#define FUTHARK_COMPAT_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- context configuration ------------------------------------------------------------------ */
struct futhark_context_config;
struct futhark_context_config *futhark_context_config_new(void);
void futhark_context_config_free(struct futhark_context_config *cfg);
void futhark_context_config_set_debugging(struct futhark_context_config *cfg, int flag);   /* accepted, no effect */
void futhark_context_config_set_profiling(struct futhark_context_config *cfg, int flag);   /* accepted, no effect */
void futhark_context_config_set_logging(struct futhark_context_config *cfg, int flag);     /* accepted, no effect */
/* GPU backends of Futhark take a device name/number here; "#k" or "k" selects HIP device k. */
void futhark_context_config_set_device(struct futhark_context_config *cfg, const char *s);

/* ---- context -> hark_context_new / _free / _sync / _get_error / _trim ------------------------- */
struct futhark_context;
struct futhark_context *futhark_context_new(struct futhark_context_config *cfg);
void futhark_context_free(struct futhark_context *ctx);
int futhark_context_sync(struct futhark_context *ctx);
char *futhark_context_get_error(struct futhark_context *ctx);      /* malloc'd, caller frees; NULL if none */
char *futhark_context_report(struct futhark_context *ctx);         /* malloc'd, caller frees */
int futhark_context_clear_caches(struct futhark_context *ctx);
void futhark_context_pause_profiling(struct futhark_context *ctx);
void futhark_context_unpause_profiling(struct futhark_context *ctx);

/* ---- arrays -> hark_table_new_2d / hark_result_values_2d / _shape / _free --------------------- */
struct futhark_i32_1d;
struct futhark_i32_1d *futhark_new_i32_1d(struct futhark_context *ctx, const int32_t *data, int64_t dim0);
int futhark_free_i32_1d(struct futhark_context *ctx, struct futhark_i32_1d *arr);
int futhark_values_i32_1d(struct futhark_context *ctx, struct futhark_i32_1d *arr, int32_t *data);
const int64_t *futhark_shape_i32_1d(struct futhark_context *ctx, struct futhark_i32_1d *arr);

struct futhark_i32_2d;
struct futhark_i32_2d *futhark_new_i32_2d(struct futhark_context *ctx, const int32_t *data, int64_t dim0, int64_t dim1);
int futhark_free_i32_2d(struct futhark_context *ctx, struct futhark_i32_2d *arr);
int futhark_values_i32_2d(struct futhark_context *ctx, struct futhark_i32_2d *arr, int32_t *data);
const int64_t *futhark_shape_i32_2d(struct futhark_context *ctx, struct futhark_i32_2d *arr);

struct futhark_u32_2d;
struct futhark_u32_2d *futhark_new_u32_2d(struct futhark_context *ctx, const uint32_t *data, int64_t dim0, int64_t dim1);
int futhark_free_u32_2d(struct futhark_context *ctx, struct futhark_u32_2d *arr);
int futhark_values_u32_2d(struct futhark_context *ctx, struct futhark_u32_2d *arr, uint32_t *data);
const int64_t *futhark_shape_u32_2d(struct futhark_context *ctx, struct futhark_u32_2d *arr);

/* ---- entry points ------------------------------------------------------------------------------ */
/* main.fut:7 -> hark_entry_query_sel */
int futhark_entry_query_sel(struct futhark_context *ctx, struct futhark_i32_2d **out0,
                            const struct futhark_i32_2d *in0, const struct futhark_i32_1d *in1);
/* main.fut:9 -> hark_entry_query_groupby */
int futhark_entry_query_groupby(struct futhark_context *ctx, struct futhark_u32_2d **out0,
                                const struct futhark_u32_2d *in0, const int32_t in1,
                                const struct futhark_i32_1d *in2, const struct futhark_i32_1d *in3);
/* join.fut:52 -> hark_entry_join */
int futhark_entry_join(struct futhark_context *ctx, struct futhark_u32_2d **out0,
                       const struct futhark_u32_2d *in0, const struct futhark_u32_2d *in1,
                       const int32_t in2, const int32_t in3,
                       const struct futhark_i32_1d *in4, const struct futhark_i32_1d *in5);

#ifdef __cplusplus
}
#endif
#endif /* FUTHARK_COMPAT_H */
