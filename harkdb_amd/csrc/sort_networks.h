// sort_networks.h -- sorting networks over N registers (compare-exchange lists: optimal 4- and 8-input networks,
// Batcher's odd-even merge sort for 16 inputs, 63 comparators), fully unrolled: no dynamic indexing, so the values stay in
// VGPRs.  `lt(a, b)` is a strict order; the networks are not stable by themselves -- callers that need stability put
// the original position into the order.  (Checked by the 0-1 principle when they were written: profiles/r02_notes.md.)
#pragma once

template <typename T, typename LT>
__device__ __forceinline__ void net_cex(T &a, T &b, LT lt) { if (lt(b, a)) { const T t = a; a = b; b = t; } }

template <typename T, typename LT>
__device__ __forceinline__ void net_sort4(T (&v)[4], LT lt)
{
#define CX(a, b) net_cex(v[a], v[b], lt)
    CX(0, 1); CX(2, 3); CX(0, 2); CX(1, 3); CX(1, 2);
#undef CX
}

template <typename T, typename LT>
__device__ __forceinline__ void net_sort8(T (&v)[8], LT lt)
{
#define CX(a, b) net_cex(v[a], v[b], lt)
    CX(0, 1); CX(2, 3); CX(4, 5); CX(6, 7);
    CX(0, 2); CX(1, 3); CX(4, 6); CX(5, 7);
    CX(1, 2); CX(5, 6); CX(0, 4); CX(3, 7);
    CX(1, 5); CX(2, 6);
    CX(1, 4); CX(3, 6);
    CX(2, 4); CX(3, 5);
    CX(3, 4);
#undef CX
}

template <typename T, typename LT>
__device__ __forceinline__ void net_sort16(T (&v)[16], LT lt)
{
#define CX(a, b) net_cex(v[a], v[b], lt)
    CX(0, 1); CX(2, 3); CX(0, 2); CX(1, 3); CX(1, 2); CX(4, 5); CX(6, 7); CX(4, 6); CX(5, 7); CX(5, 6); CX(0, 4);
    CX(2, 6); CX(2, 4); CX(1, 5); CX(3, 7); CX(3, 5); CX(1, 2); CX(3, 4); CX(5, 6); CX(8, 9); CX(10, 11);
    CX(8, 10); CX(9, 11); CX(9, 10); CX(12, 13); CX(14, 15); CX(12, 14); CX(13, 15); CX(13, 14); CX(8, 12);
    CX(10, 14); CX(10, 12); CX(9, 13); CX(11, 15); CX(11, 13); CX(9, 10); CX(11, 12); CX(13, 14); CX(0, 8);
    CX(4, 12); CX(4, 8); CX(2, 10); CX(6, 14); CX(6, 10); CX(2, 4); CX(6, 8); CX(10, 12); CX(1, 9); CX(5, 13);
    CX(5, 9); CX(3, 11); CX(7, 15); CX(7, 11); CX(3, 5); CX(7, 9); CX(11, 13); CX(1, 2); CX(3, 4); CX(5, 6);
    CX(7, 8); CX(9, 10); CX(11, 12); CX(13, 14);
#undef CX
}
