// rtn_anchor_dev.h — device-side view of rtn_anchor_cfg_t and index -> (level, cell, anchor).
// Anchor order follows anchors_for_shape (model/anchors.py:194-202, 227-236):
// level -> y -> x -> anchor (ratio-major, scale-minor).
#pragma once
#include "rtn_internal.h"

struct DevAnchorCfg {
    int nlevels, A, total;
    int H[RTN_MAX_GROUPS], W[RTN_MAX_GROUPS], stride[RTN_MAX_GROUPS];
    int off[RTN_MAX_GROUPS + 1];
    double base[RTN_MAX_GROUPS][16][4];
};

static inline int make_dev_cfg(rtn_ctx* h, const rtn_anchor_cfg_t* cfg, DevAnchorCfg* d) {
    if (!cfg) return rtn_fail(h, RTN_EINVAL, "anchors: null cfg");
    if (cfg->nlevels < 1 || cfg->nlevels > RTN_MAX_GROUPS || cfg->A < 1 || cfg->A > 16)
        return rtn_fail(h, RTN_EINVAL, "anchors: nlevels %d / A %d out of range", cfg->nlevels, cfg->A);
    memset(d, 0, sizeof(*d));
    d->nlevels = cfg->nlevels;
    d->A = cfg->A;
    long long off = 0;
    for (int l = 0; l < cfg->nlevels; ++l) {
        if (cfg->H[l] < 1 || cfg->W[l] < 1 || cfg->stride[l] < 1) return rtn_fail(h, RTN_EINVAL, "anchors: level %d empty", l);
        if (cfg->anchor_off[l] != off) return rtn_fail(h, RTN_EINVAL, "anchors: anchor_off[%d]=%d, expected %lld", l, cfg->anchor_off[l], off);
        d->H[l] = cfg->H[l]; d->W[l] = cfg->W[l]; d->stride[l] = cfg->stride[l];
        d->off[l] = (int)off;
        off += (long long)cfg->H[l] * cfg->W[l] * cfg->A;
        if (off > (1ll << 30)) return rtn_fail(h, RTN_EINVAL, "anchors: too many anchors");
        memcpy(d->base[l], cfg->base[l], sizeof(double) * 16 * 4);
    }
    if (cfg->anchor_off[cfg->nlevels] != off) return rtn_fail(h, RTN_EINVAL, "anchors: anchor_off total mismatch");
    for (int l = cfg->nlevels; l <= RTN_MAX_GROUPS; ++l) d->off[l] = (int)off;
    d->total = (int)off;
    return RTN_OK;
}

struct AnchorIdx { int level, a, x, y; };

__device__ __forceinline__ AnchorIdx locate(const DevAnchorCfg& c, int n) {
    int l = 0;
#pragma unroll
    for (int i = 1; i < RTN_MAX_GROUPS; ++i)
        if (i < c.nlevels && n >= c.off[i]) l = i;
    const int local = n - c.off[l];
    const int cell = local / c.A;
    AnchorIdx r;
    r.level = l;
    r.a = local - cell * c.A;
    r.y = cell / c.W[l];
    r.x = cell - r.y * c.W[l];
    return r;
}

