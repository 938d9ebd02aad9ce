// Compile-time index algebra of the reduced / m-primary coefficient layouts (reference
// model/EF_layers.py:1441-1474, 1514-1549).  All functions are constexpr so that fully unrolled kernels
// address their register arrays with constants.
#pragma once

template <int L, int M>
struct SO3Idx {
    static constexpr int K = (L + 1) * (L + 1);
    static constexpr int mm(int l) { return l < M ? l : M; }
    static constexpr int nr(int l) { return 2 * mm(l) + 1; }  // reduced rows of block l
    static constexpr int kr_off(int l) {
        int s = 0;
        for (int i = 0; i < l; ++i) s += nr(i);
        return s;
    }
    static constexpr int KR = kr_off(L + 1);
    static constexpr int w_off(int l) {  // offset of block l inside one edge's reduced-Wigner record
        int s = 0;
        for (int i = 0; i < l; ++i) s += nr(i) * (2 * i + 1);
        return s;
    }
    // record length in floats, rounded up to a multiple of 4 so that every edge's record starts 16-byte aligned (the
    // node-gradient kernel of k4 reads records with float4 loads; unaligned dwordx4 loads were 1.5x slower than the
    // scalar version they replaced)
    static constexpr int WSZ = (w_off(L + 1) + 3) / 4 * 4;
    static constexpr int msize(int m) { return L - m + 1; }
    static constexpr int m_off(int m) {  // first m-primary row of order +m
        if (m == 0) return 0;
        int s = L + 1;
        for (int i = 1; i < m; ++i) s += 2 * msize(i);
        return s;
    }
    static constexpr int mpos(int l, int m) {  // m-primary row of coefficient (l, m)
        return m == 0 ? l : (m > 0 ? m_off(m) + (l - m) : m_off(-m) + msize(-m) + (l + m));
    }
    static constexpr int rad_row(int l, int m) {  // row of the radial weight shared by (l, +m) and (l, -m)
        int a = m < 0 ? -m : m;
        if (a == 0) return l;
        int s = L + 1;
        for (int i = 1; i < a; ++i) s += msize(i);
        return s + (l - a);
    }
    static constexpr int RAD_ROWS = rad_row(L, M) + 1;
    // rotate_inv rescale sqrt((2l+1)/(2M+1)) for l > M (EF:1539-1547); evaluated on the host into a table.
};
