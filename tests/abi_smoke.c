/* abi_smoke.c -- a plain C caller of libhark.so: the two statements the reference itself runs
 * (README.md:42 `select col1, col3 from game_1`; test.py:7 `select col1, max(col3) ... group by col1`)
 * on data.csv, once through the hark_* ABI (include/hark.h) and once through the generated-Futhark-API
 * names (include/futhark_compat.h).  Expected outputs are goldens G1 and G2.
 * Built and run by tests/test_gpu_abi_c.py:  gcc abi_smoke.c -I include -L harkdb_amd -lhark */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "hark.h"
#include "futhark_compat.h"

static const int32_t DATA[7][8] = {   /* tests/golden/data.csv, rows 2-8 */
    {6, 6, 6, 6, 6, 6, 6, 6}, {0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0},
    {0, 0, 0, 0, 0, 0, 0, 0}, {6, 6, 6, 6, 6, 6, 6, 6}, {1, 2, 3, 4, 5, 3, 2, 1}};
static const int32_t G1[7][2] = {{6, 6}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {6, 6}, {1, 3}};
static const uint32_t G2[3][3] = {{0, 0, 0}, {1, 1, 3}, {6, 6, 6}};

#define CHECK(cond, msg) do { if (!(cond)) { fprintf(stderr, "FAIL %s (%s:%d)\n", msg, __FILE__, __LINE__); return 1; } } while (0)

static int via_hark(void)
{
    hark_context *ctx = NULL;
    CHECK(hark_context_new(&ctx, 0) == HARK_OK, "hark_context_new");
    hark_table *t = NULL;
    CHECK(hark_table_new_2d(ctx, &t, DATA, HARK_I32, 7, 8, /*row_stride*/ 8, /*col_stride*/ 1) == HARK_OK, "hark_table_new_2d");
    int32_t cols[2] = {0, 2};
    hark_result *r = NULL;
    if (hark_entry_query_sel(ctx, &r, t, cols, 2)) { puts(hark_context_get_error(ctx)); return 1; }
    int64_t rn = 0, rm = 0;
    hark_result_shape(r, &rn, &rm);
    CHECK(rn == 7 && rm == 2, "query_sel shape");
    int32_t out[7][2];
    CHECK(hark_result_values_2d(ctx, r, out, HARK_I32) == HARK_OK, "values_2d");   /* row-major [rn][rm], like futhark_values_i32_2d */
    CHECK(memcmp(out, G1, sizeof G1) == 0, "G1");
    hark_result_free(ctx, r);
    int32_t bad[1] = {8};
    CHECK(hark_entry_query_sel(ctx, &r, t, bad, 1) == HARK_EBOUNDS, "out-of-range column is a bounds error");
    CHECK(strlen(hark_context_get_error(ctx)) > 0, "error text");
    int32_t s_cols[2] = {0, 2}, t_cols[2] = {0, 3};
    CHECK(hark_entry_query_groupby(ctx, &r, t, 0, s_cols, 2, t_cols, 2) == HARK_OK, "query_groupby");
    hark_result_shape(r, &rn, &rm);
    CHECK(rn == 3 && rm == 3, "query_groupby shape");
    uint32_t gout[3][3];
    CHECK(hark_result_values_2d(ctx, r, gout, HARK_U32) == HARK_OK, "values_2d u32");
    CHECK(memcmp(gout, G2, sizeof G2) == 0, "G2");
    hark_result_free(ctx, r);
    hark_table_free(ctx, t);
    hark_context_free(ctx);
    return 0;
}

static int via_futhark_names(void)
{
    struct futhark_context_config *cfg = futhark_context_config_new();
    struct futhark_context *ctx = futhark_context_new(cfg);
    char *err = futhark_context_get_error(ctx);
    if (err) { fprintf(stderr, "FAIL futhark_context_new: %s\n", err); free(err); return 1; }
    struct futhark_i32_2d *db = futhark_new_i32_2d(ctx, &DATA[0][0], 7, 8);
    CHECK(db != NULL, "futhark_new_i32_2d");
    int32_t cols[2] = {0, 2};
    struct futhark_i32_1d *sc = futhark_new_i32_1d(ctx, cols, 2);
    struct futhark_i32_2d *res = NULL;
    CHECK(futhark_entry_query_sel(ctx, &res, db, sc) == 0, "futhark_entry_query_sel");
    CHECK(futhark_context_sync(ctx) == 0, "futhark_context_sync");
    const int64_t *shape = futhark_shape_i32_2d(ctx, res);
    CHECK(shape[0] == 7 && shape[1] == 2, "shape");
    int32_t out[7][2];
    CHECK(futhark_values_i32_2d(ctx, res, &out[0][0]) == 0, "futhark_values_i32_2d");
    CHECK(memcmp(out, G1, sizeof G1) == 0, "G1 through futhark_*");
    int32_t back[7][8];
    CHECK(futhark_values_i32_2d(ctx, db, &back[0][0]) == 0 && memcmp(back, DATA, sizeof DATA) == 0, "new -> values round trip");
    futhark_free_i32_2d(ctx, res);
    futhark_free_i32_2d(ctx, db);

    struct futhark_u32_2d *udb = futhark_new_u32_2d(ctx, (const uint32_t *)&DATA[0][0], 7, 8);
    int32_t tcols[2] = {0, 3};
    struct futhark_i32_1d *tc = futhark_new_i32_1d(ctx, tcols, 2);
    struct futhark_u32_2d *gres = NULL;
    CHECK(futhark_entry_query_groupby(ctx, &gres, udb, 0, sc, tc) == 0, "futhark_entry_query_groupby");
    shape = futhark_shape_u32_2d(ctx, gres);
    CHECK(shape[0] == 3 && shape[1] == 3, "groupby shape");
    uint32_t gout[3][3];
    CHECK(futhark_values_u32_2d(ctx, gres, &gout[0][0]) == 0, "futhark_values_u32_2d");
    CHECK(memcmp(gout, G2, sizeof G2) == 0, "G2 through futhark_*");
    int32_t bad[1] = {9};
    struct futhark_i32_1d *bc = futhark_new_i32_1d(ctx, bad, 1);
    struct futhark_u32_2d *none = NULL;
    CHECK(futhark_entry_query_groupby(ctx, &none, udb, 0, bc, tc) != 0, "bounds failure is non-zero");
    err = futhark_context_get_error(ctx);
    CHECK(err != NULL, "error string after a failing entry");
    free(err);
    futhark_free_u32_2d(ctx, gres);
    futhark_free_u32_2d(ctx, udb);
    futhark_free_i32_1d(ctx, sc); futhark_free_i32_1d(ctx, tc); futhark_free_i32_1d(ctx, bc);
    futhark_context_free(ctx);
    futhark_context_config_free(cfg);
    return 0;
}

int main(void)
{
    if (via_hark()) return 1;
    if (via_futhark_names()) return 2;
    puts("abi_smoke ok: G1 and G2 through hark_* and through futhark_*");
    return 0;
}
