/*
 * rrt_noise_plan.h -- host side of the lattice-hash tables (rrt_noise_table): which lattice points the noise3D call families
 * of densities.h:20-132 can reach within a window of the clock, the boxes (dense layout) and band boxes (banded layout) that
 * cover them, their sizes, and the NoiseTableObject registry.  Included by rrt_hip.hip inside its anonymous namespace, after
 * rrt_kernels.h (round 6: split out of rrt_hip.hip, nothing else changed).  Device side: NoiseLut / DustBands in rrt_device.h.
 */
#ifndef RRT_NOISE_PLAN_H
#define RRT_NOISE_PLAN_H

/* ------------------------------------------------------------------ lattice-hash tables (rrt_noise_table)
 * Two dense boxes of the integer lattice, one for the accretion fbm and one for the dust-cloud noise calls
 * (layout and use: NoiseLut in rrt_device.h).  The boxes are computed on the host from the coordinate ranges
 * those calls can reach for t0 <= time <= t1 (lut_boxes below); a launch with a time outside that window
 * simply runs the arithmetic kernels. */
constexpr int kMaxBands = 64;
/* BANDED layout (round 5; rrt_device.h: DustBands): the three fine dust families in one small box per omega band */
struct BandPlan {
    int n_bands;                                   /* 0: dense layout */
    float w_min, w_scale;
    bool present[rrt::kBandFamilies];              /* family in the table's coverage */
    LutBox box[rrt::kBandFamilies][kMaxBands];
    unsigned cell0[rrt::kBandFamilies][kMaxBands]; /* first cell of the box, in cells from d_cells */
    LutBox acc_box[rrt::kLutAccOctaves];           /* the accretion table, one box per octave (octave 0 first in the allocation) */
    unsigned acc_cell0[rrt::kLutAccOctaves];
    unsigned dust_cell0;                           /* the coarse dust families' box */
    size_t entries_offset;                         /* byte offset of the BandLut array inside the allocation */
};
struct NoiseTableObject {
    float4* d_cells;          /* dense: accretion box, dust box.  banded: accretion octave boxes, the coarse dust families' box, the band boxes, the BandLut records */
    bool banded;
    BandPlan bands;
    size_t bytes;
    float t0, t1;             /* launches with t0 <= time <= t1 read the table */
    int coverage;             /* RRT_TABLE_FULL / _COARSE / _COARSEST */
    unsigned acc_families, dust_families;
    LutBox acc, dust;
    int device;
};
std::mutex g_nt_mu;
std::unordered_map<int, NoiseTableObject> g_nt;
int g_nt_next = 1;

struct Interval {
    double lo, hi;
    Interval scaled(double s) const { return s >= 0 ? Interval{lo * s, hi * s} : Interval{hi * s, lo * s}; }
    Interval shifted(double a, double b) const { return Interval{lo + a, hi + b}; }      /* + [a, b] */
    Interval widened(double w) const { return Interval{lo - w, hi + w}; }
};
struct Reach {                 /* running union of the lattice points a noise3D call family can touch */
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    void add(const Interval c[3]) {
        for (int k = 0; k < 3; ++k) { lo[k] = std::fmin(lo[k], c[k].lo); hi[k] = std::fmax(hi[k], c[k].hi); }
    }
    /* `octaves` octaves of fbm starting at c: p -> p*2.05 + 10 (math_utils.h:116) */
    void add_fbm(Interval c[3], int octaves) {
        for (int o = 0; o < octaves; ++o) {
            add(c);
            for (int k = 0; k < 3; ++k) c[k] = c[k].scaled(2.05).shifted(10.0, 10.0);
        }
    }
    LutBox box() const {       /* floor(lo) .. floor(hi) + 1 are the corners used; two cells of slack on every side */
        LutBox b;
        int l[3], h[3];
        for (int k = 0; k < 3; ++k) { l[k] = (int)std::floor(lo[k]) - 2; h[k] = (int)std::floor(hi[k]) + 1 + 2; }
        b.x0 = l[0]; b.y0 = l[1]; b.z0 = l[2];
        b.nx = h[0] - l[0] + 1; b.ny = h[1] - l[1] + 1; b.nz = h[2] - l[2] + 1;
        return b;
    }
};

/* Which noise3D call families a table of a given coverage serves (bit layout = the `from_table` words of
 * accretion_density_at / dust_density_at in rrt_device.h).  The finest families dominate the volume of the dust box
 * (it grows with the cube of the scale: 4.41^3 = 86 against 2.1^3 = 9 and 1), so a coarser coverage buys a table an
 * order of magnitude smaller for sequences that run long. */
void coverage_families(int coverage, unsigned& acc, unsigned& dust) {
    acc = (1u << rrt::kLutAccOctaves) - 1u;
    dust = 0xfu | (((1u << rrt::kLutRidgeOctaves) - 1u) << 4) | (rrt::kLutDetail ? 256u : 0u);
    if (coverage >= RRT_TABLE_COARSE) dust &= ~((1u << 6) | 256u);              /* without ridge octave 2 (4.41 cells per unit) and the detail octave (4.0) */
    if (coverage >= RRT_TABLE_COARSEST) { dust &= ~(1u << 5); acc &= 0x7u; }    /* without ridge octave 1 (2.1) and accretion octave 3 (3.9) */
}

/* Coordinate ranges of the table-served noise calls for t0 <= time <= t1, from the constants of
 * densities.h (every bound is taken generously: the functions only run for rc in [10, 25], the
 * accretion one for |y| < 4 and the dust one for |y| < 0.75 -- the zone tests of raymarcher.cu:57-58 --
 * |sin|, |cos| <= 1 + 1e-6, |atan2| <= pi + 1e-6, |noise3D| < 1 + 1e-6, hence |fbm(.,2)| < 0.76).
 * The dust box does NOT stay bounded for a window that slides: its z coordinate is 10 (azimuth - time * omega)
 * with omega = (10/rc)^1.5 in [0.253, 1] (densities.h:88-93) -- differential rotation -- so the reachable z range
 * is 10 [-(pi + max t omega), pi - min t omega]: its width grows like 0.75 t0 + (t1 - t0).  A window bounds it from
 * both sides, a coarser coverage cuts the scale factor. */
void lut_boxes(double t0, double t1, unsigned acc_fam, unsigned dust_fam, LutBox& acc, LutBox& dust) {
    const double pi = 3.14159265358979 + 1e-5;
    const double slack = 1e-6 * (std::fabs(t0) + std::fabs(t1)) + 1e-3;      /* float rounding of time * rate at large times */
    {   /* getAccretionDensity, densities.h:44-54: (rc cos, 4y, rc sin)*0.45 + (0, 0.35 t, 0) */
        Reach r;
        Interval c[3] = {Interval{-25.0, 25.0}.scaled(0.45).widened(1e-3),
                         Interval{-16.0, 16.0}.scaled(0.45).shifted(0.35 * t0, 0.35 * t1).widened(slack),
                         Interval{-25.0, 25.0}.scaled(0.45).widened(1e-3)};
        int octaves = 0;
        while (octaves < rrt::kLutAccOctaves && ((acc_fam >> octaves) & 1u)) ++octaves;
        if (octaves == 0) octaves = 1;                         /* never an empty box */
        r.add_fbm(c, octaves);
        acc = r.box();
    }
    {   /* getDustCloudDensity, densities.h:93: coords = (0.8 rc, 15 y, 10 (phi - t*omega)), omega in [0.25, 1] */
        Reach r;
        const double w_lo = 0.25, w_hi = 1.0;
        const double tw_max = t1 >= 0.0 ? t1 * w_hi : t1 * w_lo;          /* max of t * omega over the window */
        const double tw_min = t0 >= 0.0 ? t0 * w_lo : t0 * w_hi;          /* min */
        const Interval sc[3] = {Interval{8.0, 20.0}.widened(1e-3), Interval{-11.25, 11.25}.widened(1e-3),
                                Interval{-(pi + tw_max) * 10.0, (pi - tw_min) * 10.0}.widened(1e-2 + 10.0 * slack)};
        const double off1[3][3] = {{0, 0, 0}, {1, 2, 3}, {4, 5, 6}}, off2[3][3] = {{0, 0, 0}, {2, 1, 0}, {0, 3, 1}};
        const int w1_oct = (dust_fam & 2u) ? 2 : 1, w2_oct = (dust_fam & 8u) ? 2 : 1;
        for (int k = 0; k < 3; ++k) {                          /* :95-99 */
            Interval c[3];
            for (int ax = 0; ax < 3; ++ax) c[ax] = sc[ax].scaled(0.15).shifted(off1[k][ax], off1[k][ax]);
            r.add_fbm(c, w1_oct);
        }
        if (dust_fam & 4u) for (int k = 0; k < 3; ++k) {       /* :101-106: (coords + 3 w1)*0.4 + offsets */
            Interval c[3];
            for (int ax = 0; ax < 3; ++ax) c[ax] = sc[ax].widened(3.0 * 0.76).scaled(0.4).shifted(off2[k][ax], off2[k][ax]);
            r.add_fbm(c, w2_oct);
        }
        double freq = 1.0;
        for (int k = 0; k < rrt::kLutRidgeOctaves; ++k) {      /* :111-120: (coords + 1.5 w2)*freq */
            if ((dust_fam >> (4 + k)) & 1u) {
                Interval c[3];
                for (int ax = 0; ax < 3; ++ax) c[ax] = sc[ax].widened(1.5 * 0.76).scaled(freq);
                r.add(c);
            }
            freq *= 2.1;
        }
        if (rrt::kLutDetail && (dust_fam & 256u)) {            /* :127: (coords + 1.5 w2)*4 + (0, 0.5 t, 0), first octave only */
            Interval c[3];
            for (int ax = 0; ax < 3; ++ax) c[ax] = sc[ax].widened(1.5 * 0.76).scaled(4.0);
            c[1] = c[1].shifted(0.5 * t0, 0.5 * t1).widened(slack);
            r.add_fbm(c, 1);
        }
        dust = r.box();
    }
}

/* the accretion table's octave `o` alone (banded layout): lut_boxes' first block, one octave instead of their union */
LutBox acc_octave_box(double t0, double t1, int o) {
    const double slack = 1e-6 * (std::fabs(t0) + std::fabs(t1)) + 1e-3;
    Interval c[3] = {Interval{-25.0, 25.0}.scaled(0.45).widened(1e-3),
                     Interval{-16.0, 16.0}.scaled(0.45).shifted(0.35 * t0, 0.35 * t1).widened(slack),
                     Interval{-25.0, 25.0}.scaled(0.45).widened(1e-3)};
    for (int k = 0; k < o; ++k)
        for (int ax = 0; ax < 3; ++ax) c[ax] = c[ax].scaled(2.05).shifted(10.0, 10.0);
    Reach r;
    r.add(c);
    return r.box();
}

/* noise3d_lut multiplies with 24-bit operands and addresses records with 32-bit byte offsets */
bool lut_box_addressable(const LutBox& b) {
    const size_t n = (size_t)b.nx * b.ny * b.nz;
    return (size_t)b.nx * b.ny < ((size_t)1 << 23) && n < ((size_t)1 << 28) && b.nz < (1 << 23);
}

/* The fine dust families (bits of the `from_table` word: ridge octave 1, ridge octave 2, detail octave 0), their scales, and
 * the box of one of them for samples whose omega = (10/rc)^1.5 lies in [wa, wb]: the same interval arithmetic as lut_boxes,
 * with the radius and the shear restricted to the band. */
constexpr unsigned kFineDustBits[rrt::kBandFamilies] = {1u << 5, 1u << 6, 256u};
constexpr unsigned kFineDustMask = (1u << 5) | (1u << 6) | 256u;
LutBox band_box(int family, double t0, double t1, double wa, double wb) {
    const double pi = 3.14159265358979 + 1e-5;
    const double slack = 1e-6 * (std::fabs(t0) + std::fabs(t1)) + 1e-3;
    wa = std::fmax(wa, 0.25); wb = std::fmin(wb, 1.0);
    const double rc_lo = std::fmax(10.0, 10.0 / std::pow(wb, 2.0 / 3.0)) - 1e-3, rc_hi = std::fmin(25.0, 10.0 / std::pow(wa, 2.0 / 3.0)) + 1e-3;
    const double tw[4] = {t0 * wa, t0 * wb, t1 * wa, t1 * wb};
    const double tw_min = std::fmin(std::fmin(tw[0], tw[1]), std::fmin(tw[2], tw[3])), tw_max = std::fmax(std::fmax(tw[0], tw[1]), std::fmax(tw[2], tw[3]));
    const Interval sc[3] = {Interval{0.8 * rc_lo, 0.8 * rc_hi}.widened(1e-3), Interval{-11.25, 11.25}.widened(1e-3),
                            Interval{-(pi + tw_max) * 10.0, (pi - tw_min) * 10.0}.widened(1e-2 + 10.0 * slack)};
    const double freq = family == 0 ? 2.1 : (family == 1 ? 2.1 * 2.1 : 4.0);
    Interval c[3];
    for (int ax = 0; ax < 3; ++ax) c[ax] = sc[ax].widened(1.5 * 0.76).scaled(freq);       /* (coords + 1.5 w2) * freq, densities.h:108-128 */
    if (family == 2) c[1] = c[1].shifted(0.5 * t0, 0.5 * t1).widened(slack);               /* + (0, 0.5 t, 0) */
    Reach r;
    r.add(c);
    return r.box();
}

/* the banded plan with the fewest bytes over n_bands in {1, 2, 4 ... kMaxBands}; false if no band count is addressable */
bool plan_bands(double t0, double t1, unsigned dust_fam, BandPlan& bp, size_t& band_cells) {
    const double w_min = 0.2529, w_max = 1.0001;          /* omega = (10/rc)^1.5 for rc in [10, 25]: [0.25298, 1] */
    bool found = false;
    for (int nb = 1; nb <= kMaxBands; nb *= 2) {
        BandPlan cand;
        memset(&cand, 0, sizeof(cand));
        cand.n_bands = nb; cand.w_min = (float)w_min; cand.w_scale = (float)(nb / (w_max - w_min));
        size_t cells = 0;
        bool ok = true;
        for (int f = 0; f < rrt::kBandFamilies && ok; ++f) {
            cand.present[f] = (dust_fam & kFineDustBits[f]) != 0;
            for (int b = 0; b < nb && ok; ++b) {
                /* the device picks the band as (int)((omega - w_min) * w_scale) in binary32: 1e-5 of slack on either side */
                const double wa = (double)cand.w_min + (double)b / (double)cand.w_scale - 1e-5, wb = (double)cand.w_min + (double)(b + 1) / (double)cand.w_scale + 1e-5;
                LutBox bx = cand.present[f] ? band_box(f, t0, t1, b == 0 ? 0.25 : wa, b == nb - 1 ? 1.0 : wb) : LutBox{0, 0, 0, 2, 2, 2};
                ok = lut_box_addressable(bx);
                cand.box[f][b] = bx;
                cells += (size_t)bx.nx * bx.ny * bx.nz;
            }
        }
        if (ok && (!found || cells < band_cells)) { bp = cand; band_cells = cells; found = true; }
    }
    return found;
}

/* boxes + byte size of a table over [t0, t1] at `coverage` (RRT_TABLE_FULL .. COARSEST, optionally | RRT_TABLE_BANDED or
 * | RRT_TABLE_DENSE to force a layout); RRT_ERR_INVALID_ARGUMENT for what create would refuse.  Without a forced layout:
 * dense (one box for all dust families, the layout every window near the origin of the clock gets, no extra loads) unless
 * that box is unaddressable or larger than kDenseLimitBytes and the banded plan is smaller. */
constexpr size_t kDenseLimitBytes = (size_t)768 << 20;
int plan_table(float t0, float t1, int coverage_arg, NoiseTableObject& nt) {
    if (!(t0 <= t1) || !(t0 >= -1.0e4f) || !(t1 <= 1.0e4f)) return RRT_ERR_INVALID_ARGUMENT;
    const int coverage = coverage_arg & 0xf, forced = coverage_arg & ~0xf;
    if (coverage < RRT_TABLE_FULL || coverage > RRT_TABLE_COARSEST) return RRT_ERR_INVALID_ARGUMENT;
    if (forced != 0 && forced != RRT_TABLE_BANDED && forced != RRT_TABLE_DENSE) return RRT_ERR_INVALID_ARGUMENT;
    memset(&nt, 0, sizeof(nt));
    nt.t0 = t0; nt.t1 = t1; nt.coverage = coverage; nt.device = -1;
    coverage_families(coverage, nt.acc_families, nt.dust_families);
    /* dense */
    NoiseTableObject dense = nt;
    lut_boxes((double)t0, (double)t1, dense.acc_families, dense.dust_families, dense.acc, dense.dust);
    const bool dense_ok = lut_box_addressable(dense.acc) && lut_box_addressable(dense.dust);
    dense.bytes = ((size_t)dense.acc.nx * dense.acc.ny * dense.acc.nz + (size_t)dense.dust.nx * dense.dust.ny * dense.dust.nz) * sizeof(float4);
    /* banded: the coarse dust families keep the one box, the fine ones (if the coverage has any) get a box per band */
    NoiseTableObject band = nt;
    bool band_ok = (nt.dust_families & kFineDustMask) != 0 && forced != RRT_TABLE_DENSE;
    if (band_ok) {
        LutBox acc_union;
        lut_boxes((double)t0, (double)t1, band.acc_families, band.dust_families & ~kFineDustMask, acc_union, band.dust);
        size_t band_cells = 0;
        band_ok = lut_box_addressable(band.dust) && plan_bands((double)t0, (double)t1, band.dust_families, band.bands, band_cells);
        if (band_ok) {
            size_t at = 0;
            for (int o = 0; o < rrt::kLutAccOctaves; ++o) {          /* octaves the coverage does not serve: a token box, never read */
                const LutBox bx = ((band.acc_families >> o) & 1u) || o == 0 ? acc_octave_box((double)t0, (double)t1, o) : LutBox{0, 0, 0, 2, 2, 2};
                band_ok = band_ok && lut_box_addressable(bx);
                band.bands.acc_box[o] = bx; band.bands.acc_cell0[o] = (unsigned)at;
                at += (size_t)bx.nx * bx.ny * bx.nz;
            }
            band.acc = band.bands.acc_box[0];                          /* what rrt_noise_table_info reports as "the" accretion box */
            band.bands.dust_cell0 = (unsigned)at;
            at += (size_t)band.dust.nx * band.dust.ny * band.dust.nz;
            for (int f = 0; f < rrt::kBandFamilies; ++f)
                for (int b = 0; b < band.bands.n_bands; ++b) {
                    band.bands.cell0[f][b] = (unsigned)at;
                    at += (size_t)band.bands.box[f][b].nx * band.bands.box[f][b].ny * band.bands.box[f][b].nz;
                }
            band_ok = at < ((size_t)1 << 32);
            band.bands.entries_offset = (at * sizeof(float4) + 255) & ~(size_t)255;
            band.bytes = band.bands.entries_offset + ((size_t)rrt::kBandFamilies * band.bands.n_bands + rrt::kLutAccOctaves) * sizeof(rrt::BandLut);
            band.banded = true;
        }
    }
    const bool take_band = band_ok && (forced == RRT_TABLE_BANDED || !dense_ok || (dense.bytes > kDenseLimitBytes && band.bytes < dense.bytes));
    if (forced == RRT_TABLE_BANDED && !band_ok) return RRT_ERR_INVALID_ARGUMENT;
    if (take_band) { nt = band; return RRT_OK; }
    if (!dense_ok) return RRT_ERR_INVALID_ARGUMENT;
    nt = dense;
    return RRT_OK;
}

/* first cell of the (coarse or only) dust box */
size_t dust_cell0(const NoiseTableObject& nt) { return nt.banded ? (size_t)nt.bands.dust_cell0 : (size_t)nt.acc.nx * nt.acc.ny * nt.acc.nz; }

rrt::DustBands make_bands(const NoiseTableObject& nt) {
    rrt::DustBands d;
    d.entries = nt.banded ? reinterpret_cast<const rrt::BandLut*>(reinterpret_cast<const char*>(nt.d_cells) + nt.bands.entries_offset) : nullptr;
    d.n_bands = nt.banded ? nt.bands.n_bands : 1;
    d.w_min = nt.bands.w_min; d.w_scale = nt.bands.w_scale;
    return d;
}

NoiseLut make_lut(const float4* cells, const LutBox& b, unsigned families) {
    NoiseLut L;
    L.cells = cells;
    L.families = families;
    L.nx = b.nx; L.nxy = b.nx * b.ny;
    L.origin = (b.z0 * b.ny + b.y0) * b.nx + b.x0;
    L.last = (unsigned)((size_t)b.nx * b.ny * b.nz - (size_t)L.nxy - 1);
    return L;
}

#endif /* RRT_NOISE_PLAN_H */
