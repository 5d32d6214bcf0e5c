// Temporary: entries declared in hark.h whose kernels land in the next commits.
#include "hark_internal.h"
extern "C" {
int hark_entry_query_groupby(hark_context *ctx, hark_result **out, const hark_table *, int32_t, const int32_t *, int64_t, const int32_t *, int64_t)
{ if (out) *out = nullptr; return hark_fail(ctx, HARK_EUNSUPPORTED, "query_groupby: not built yet"); }
int hark_entry_join(hark_context *ctx, hark_result **out, const hark_table *, const hark_table *, int32_t, int32_t, const int32_t *, int64_t, const int32_t *, int64_t)
{ if (out) *out = nullptr; return hark_fail(ctx, HARK_EUNSUPPORTED, "join: not built yet"); }
int hark_entry_filter_groupby(hark_context *ctx, hark_result **out, const hark_table *, int32_t, int32_t, const void *, int32_t, const int32_t *, const int32_t *, int64_t)
{ if (out) *out = nullptr; return hark_fail(ctx, HARK_EUNSUPPORTED, "filter_groupby: not built yet"); }
int hark_entry_sort(hark_context *ctx, hark_result **out, const hark_table *, int32_t, int32_t, const int32_t *, int64_t)
{ if (out) *out = nullptr; return hark_fail(ctx, HARK_EUNSUPPORTED, "sort: not built yet"); }
}
