#include "orb_host.h"

#include <algorithm>
#include <cmath>

namespace msorb {

int round_half_even(float v) { return (int)lrintf(v); }
int round_half_even(double v) { return (int)lrint(v); }
static inline int floor_int(double v) { int i = (int)v; return i - (i > v); }
static inline int ceil_int(double v) { int i = (int)v; return i + (i < v); }

void OrbParams::init(int nf, float sf, int nl, int ini, int mn) {
    nfeatures = nf; nlevels = nl; ini_th = ini; min_th = mn; scale_factor_f = sf;
    const double sfd = sf;  // the reference stores scaleFactor as double (ORBextractor.h:94)
    scale.assign(nl, 1.0f); sigma2.assign(nl, 1.0f);
    inv_scale.assign(nl, 1.0f); inv_sigma2.assign(nl, 1.0f);
    for (int i = 1; i < nl; i++) {
        scale[i] = (float)(scale[i - 1] * sfd);
        sigma2[i] = scale[i] * scale[i];
    }
    for (int i = 0; i < nl; i++) {
        inv_scale[i] = 1.0f / scale[i];
        inv_sigma2[i] = 1.0f / sigma2[i];
    }
    per_level.assign(nl, 0);
    const float factor = (float)(1.0f / sfd);
    float desired = nf * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
    int sum = 0;
    for (int l = 0; l < nl - 1; l++) {
        per_level[l] = round_half_even(desired);
        sum += per_level[l];
        desired *= factor;
    }
    per_level[nl - 1] = std::max(nf - sum, 0);

    // circular patch row extents (ORBextractor.cc:453-468)
    const int vmax = floor_int(kHalfPatch * std::sqrt(2.f) / 2 + 1);
    const int vmin = ceil_int(kHalfPatch * std::sqrt(2.f) / 2);
    const double hp2 = kHalfPatch * kHalfPatch;
    for (int v = 0; v <= vmax; ++v) umax[v] = round_half_even(std::sqrt(hp2 - v * v));
    for (int v = kHalfPatch, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

std::vector<ResizeTap> make_resize_taps(int dst_len, int src_len, bool horizontal) {
    std::vector<ResizeTap> t(dst_len);
    const double scale = 1. / ((double)dst_len / src_len);
    for (int d = 0; d < dst_len; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = floor_int(f);
        f -= s;
        if (horizontal) {  // x taps: clamp index and zero the fraction
            if (s < 0) { f = 0; s = 0; }
            if (s >= src_len - 1) { f = 0; s = src_len - 1; }
        }
        auto sat = [](float v) {
            int i = round_half_even(v);
            return (int16_t)std::min(32767, std::max(-32768, i));
        };
        ResizeTap tap;
        tap.c0 = sat((1.f - f) * 2048);
        tap.c1 = sat(f * 2048);
        auto clip = [&](int i) { return i < 0 ? 0 : (i < src_len ? i : src_len - 1); };
        tap.i0 = (int16_t)clip(s);
        tap.i1 = (int16_t)clip(s + 1);  // weights are kept when the index is clipped (y); c1==0 there for x
        t[d] = tap;
    }
    return t;
}

bool FrameGeom::build(const OrbParams& p, int rows_, int cols_) {
    rows = rows_; cols = cols_; nlevels = p.nlevels;
    cells.clear();
    size_t off = 0, boff = 0;
    int slot = 0;
    for (int l = 0; l < nlevels; l++) {
        LevelGeom& g = lv[l];
        const float s = p.inv_scale[l];
        g.w = round_half_even((float)cols * s);
        g.h = round_half_even((float)rows * s);
        if (g.w > 32000 || g.h > 32000) return false;
        g.pitch = (g.w + 63) & ~63;
        g.quota = p.per_level[l];
        g.plane_off = off;
        const size_t plane = (size_t)g.pitch * g.h;
        if (l == 0) plane0_bytes = plane;
        off += plane;
        g.blur_off = boff;
        boff += (size_t)g.pitch * ((g.h + 7) & ~7);
        // cell grid (ORBextractor.cc:789-803)
        g.min_x = kMinBorder; g.min_y = kMinBorder;
        g.max_x = g.w - kEdgeThreshold + 3;
        g.max_y = g.h - kEdgeThreshold + 3;
        const float width = (float)(g.max_x - g.min_x), height = (float)(g.max_y - g.min_y);
        if (width <= 0 || height <= 0) return false;
        g.n_cols = (int)(width / 35.f);
        g.n_rows = (int)(height / 35.f);
        if (g.n_cols < 1 || g.n_rows < 1) return false;
        g.w_cell = (int)std::ceil(width / g.n_cols);
        g.h_cell = (int)std::ceil(height / g.n_rows);
        if ((int)std::round(width / height) < 1) return false;  // nIni == 0 (ORBextractor.cc:559)
        g.cell_begin = (int)cells.size();
        for (int i = 0; i < g.n_rows; i++) {
            const float iniY = (float)(g.min_y + i * g.h_cell);
            float maxY = iniY + g.h_cell + 6;
            if (iniY >= g.max_y - 3) continue;
            if (maxY > g.max_y) maxY = (float)g.max_y;
            for (int j = 0; j < g.n_cols; j++) {
                const float iniX = (float)(g.min_x + j * g.w_cell);
                float maxX = iniX + g.w_cell + 6;
                if (iniX >= g.max_x - 6) continue;
                if (maxX > g.max_x) maxX = (float)g.max_x;
                CellDesc c;
                c.level = (int16_t)l;
                c.x0 = (int16_t)iniX; c.y0 = (int16_t)iniY;
                c.rw = (int16_t)((int)maxX - (int)iniX);
                c.rh = (int16_t)((int)maxY - (int)iniY);
                const int dw = std::max(c.rw - 6, 0), dh = std::max(c.rh - 6, 0);
                c.slot_off = slot;
                c.slot_cap = ((dw + 1) / 2) * ((dh + 1) / 2);  // strict 3x3 NMS: no two survivors are 8-adjacent
                slot += c.slot_cap;
                {
                    const int ga = c.x0 & ~3, x_lo = c.x0 + 3, x_hi = c.x0 + c.rw - 3, gx0 = x_lo & ~3;
                    const int G = std::max((x_hi - gx0 + 3) >> 2, 1), ndw = std::max((c.x0 + c.rw - ga + 3) >> 2, 1);
                    c.G = (int16_t)G;
                    const int s128 = std::max(128 / G, 1), s256 = std::max(256 / G, 1);
                    c.R128 = (int8_t)std::min((dh + s128 - 1) / s128, 127);
                    c.R256 = (int8_t)std::min((dh + s256 - 1) / s256, 127);
                    c.ndw = (int16_t)ndw;
                    c.g_magic = ((1u << 20) + G - 1) / G;
                    c.ndw_magic = ((1u << 20) + ndw - 1) / ndw;
                    c.rw_magic = ((1u << 20) + std::max<int>(c.rw, 1) - 1) / std::max<int>(c.rw, 1);
                    // rows per wave (see CellDesc): iterations of the quick-test loop = max(rows, 3) per wave
                    const int spw = std::max(64 / G, 1);
                    c.spw = (uint8_t)std::min(spw, 255);
                    c.pad_ = 0;
                    auto split = [&](int waves, int r_uniform, int max_r, uint32_t& rw_out, uint32_t& yw_out) -> uint8_t {
                        const int S = waves * spw, base = dh / S, rem = dh - base * S, plus = (rem + spw - 1) / spw;
                        int cost = 0, y = 0, worst = 0;
                        rw_out = 0; yw_out = 0;
                        for (int w = 0; w < waves; w++) {
                            const int r = base + (w < plus ? 1 : 0);
                            rw_out |= (uint32_t)r << (8 * w);
                            yw_out |= (uint32_t)std::min(y, dh) << (8 * w);
                            y += spw * r;
                            cost += std::max(r, 3);
                            worst = std::max(worst, r);
                        }
                        return (G <= 64 && dh <= 255 && worst <= max_r && cost < waves * std::max(r_uniform, 3)) ? 1 : 0;
                    };
                    c.by_wave[0] = split(2, c.R128, 5, c.rw128, c.yw128);   // GeoSmall::kMaxR
                    c.by_wave[1] = split(4, c.R256, 6, c.rw256, c.yw256);   // GeoLarge::kMaxR
                }
                cells.push_back(c);
            }
        }
        g.cell_count = (int)cells.size() - g.cell_begin;
    }
    pyramid_bytes = off;
    blur_bytes = boff;
    slots_per_image = slot;
    return true;
}

// ------------------------------------------------------------------------------------------------
// Quadtree selection.  Same observable behaviour as the reference's std::list<ExtractorNode> walk
// (ORBextractor.cc:555-779) — node order, split order, the (count, UL.x) sort with libstdc++'s
// std::sort tie behaviour, "first strictly greater response wins" — on an index-based node pool:
// nodes live in one vector and are chained by prev/next indices (push_front / erase), and a node's
// keypoints are a contiguous run of candidate indices in an append-only arena.
// ------------------------------------------------------------------------------------------------
namespace {
struct QNode {
    int x0, x1, y0, y1;  // UL=(x0,y0) UR=(x1,y0) BL=(x0,y1) BR=(x1,y1)
    int begin, count;    // run in the arena
    int prev, next;      // list links (-1 = none)
    bool no_more;
};

struct QTree {
    std::vector<QNode> pool;
    std::vector<int> arena;
    int head = -1, tail = -1, size = 0;

    void push_back(int id) {
        pool[id].prev = tail; pool[id].next = -1;
        if (tail >= 0) pool[tail].next = id; else head = id;
        tail = id; size++;
    }
    void push_front(int id) {
        pool[id].next = head; pool[id].prev = -1;
        if (head >= 0) pool[head].prev = id; else tail = id;
        head = id; size++;
    }
    int erase(int id) {  // returns the following node
        const int p = pool[id].prev, n = pool[id].next;
        if (p >= 0) pool[p].next = n; else head = n;
        if (n >= 0) pool[n].prev = p; else tail = p;
        size--;
        return n;
    }
};
}  // namespace

void distribute_quadtree(const Cand16* c, int n, int min_x, int max_x, int min_y, int max_y, int N,
                         std::vector<int>& kept) {
    kept.clear();
    if (n <= 0) return;
    const int n_ini = (int)std::round(static_cast<float>(max_x - min_x) / (max_y - min_y));
    const float hX = static_cast<float>(max_x - min_x) / n_ini;

    QTree t;
    t.pool.reserve(4 * (size_t)std::max(N, 64) + 8 * n_ini + 64);
    t.arena.reserve((size_t)n * 4);

    // initial column nodes: count, then fill (stable, candidate order kept)
    std::vector<int> col(n), cnt(n_ini, 0);
    for (int i = 0; i < n; i++) {
        col[i] = (int)((float)c[i].x / hX);
        cnt[col[i]]++;
    }
    std::vector<int> fill(n_ini, 0);
    {
        int off = 0;
        for (int i = 0; i < n_ini; i++) { fill[i] = off; off += cnt[i]; }
        t.arena.resize(n);
        std::vector<int> cur = fill;
        for (int i = 0; i < n; i++) t.arena[cur[col[i]]++] = i;
    }
    for (int i = 0; i < n_ini; i++) {
        if (cnt[i] == 0) continue;  // empty initial nodes are erased (ORBextractor.cc:597-598)
        QNode q;
        q.x0 = (int)(hX * static_cast<float>(i));
        q.x1 = (int)(hX * static_cast<float>(i + 1));
        q.y0 = 0; q.y1 = max_y - min_y;
        q.begin = fill[i]; q.count = cnt[i];
        q.no_more = (cnt[i] == 1);
        q.prev = q.next = -1;
        t.pool.push_back(q);
        t.push_back((int)t.pool.size() - 1);
    }

    struct Sized { int count, id; };
    std::vector<Sized> sized, prev_sized;
    auto less_sized = [&t](const Sized& a, const Sized& b) {  // compareNodes, ORBextractor.cc:538-553
        if (a.count < b.count) return true;
        if (a.count > b.count) return false;
        return t.pool[a.id].x0 < t.pool[b.id].x0;
    };

    // DivideNode (ORBextractor.cc:480-536) + the four push_front blocks (:640-675 / :705-740)
    auto split = [&](int id, int* n_to_expand) {
        const QNode q = t.pool[id];
        const int hx = (int)std::ceil(static_cast<float>(q.x1 - q.x0) / 2);
        const int hy = (int)std::ceil(static_cast<float>(q.y1 - q.y0) / 2);
        const int mx = q.x0 + hx, my = q.y0 + hy;
        int cc[4] = {0, 0, 0, 0};
        const size_t base = t.arena.size();
        t.arena.resize(base + q.count);
        // quadrant per key: 0=n1 (left,top) 1=n2 (right,top) 2=n3 (left,bottom) 3=n4 (right,bottom)
        int* quad = t.arena.data() + base;  // temporarily holds quadrants
        for (int k = 0; k < q.count; k++) {
            const Cand16& p = c[t.arena[q.begin + k]];
            const int qd = ((float)p.x < (float)mx ? 0 : 1) + ((float)p.y < (float)my ? 0 : 2);
            quad[k] = qd;
            cc[qd]++;
        }
        int start[4] = {0, cc[0], cc[0] + cc[1], cc[0] + cc[1] + cc[2]};
        int cur[4] = {start[0], start[1], start[2], start[3]};
        std::vector<int>& scratch = col;  // reuse
        if ((int)scratch.size() < q.count) scratch.resize(q.count);
        for (int k = 0; k < q.count; k++) scratch[cur[quad[k]]++] = t.arena[q.begin + k];
        for (int k = 0; k < q.count; k++) t.arena[base + k] = scratch[k];
        const int bx0[4] = {q.x0, mx, q.x0, mx}, bx1[4] = {mx, q.x1, mx, q.x1};
        const int by0[4] = {q.y0, q.y0, my, my}, by1[4] = {my, my, q.y1, q.y1};
        for (int ch = 0; ch < 4; ch++) {
            if (cc[ch] == 0) continue;
            QNode nn;
            nn.x0 = bx0[ch]; nn.x1 = bx1[ch]; nn.y0 = by0[ch]; nn.y1 = by1[ch];
            nn.begin = (int)base + start[ch]; nn.count = cc[ch];
            nn.no_more = (cc[ch] == 1);
            nn.prev = nn.next = -1;
            t.pool.push_back(nn);
            const int nid = (int)t.pool.size() - 1;
            t.push_front(nid);
            if (cc[ch] > 1) {
                if (n_to_expand) (*n_to_expand)++;
                sized.push_back({cc[ch], nid});
            }
        }
    };

    bool finish = false;
    while (!finish) {
        int prev_size = t.size;
        int n_to_expand = 0;
        sized.clear();
        for (int id = t.head; id >= 0;) {
            if (t.pool[id].no_more) { id = t.pool[id].next; continue; }
            split(id, &n_to_expand);
            id = t.erase(id);
        }
        if (t.size >= N || t.size == prev_size) {
            finish = true;
        } else if (t.size + n_to_expand * 3 > N) {
            while (!finish) {
                prev_size = t.size;
                prev_sized = sized;
                sized.clear();
                std::sort(prev_sized.begin(), prev_sized.end(), less_sized);
                for (int j = (int)prev_sized.size() - 1; j >= 0; j--) {
                    split(prev_sized[j].id, nullptr);
                    t.erase(prev_sized[j].id);
                    if (t.size >= N) break;
                }
                if (t.size >= N || t.size == prev_size) finish = true;
            }
        }
    }

    kept.reserve(t.size);
    for (int id = t.head; id >= 0; id = t.pool[id].next) {  // ORBextractor.cc:757-776
        const QNode& q = t.pool[id];
        int best = t.arena[q.begin];
        for (int k = 1; k < q.count; k++) {
            const int i = t.arena[q.begin + k];
            if (c[i].score > c[best].score) best = i;
        }
        kept.push_back(best);
    }
}

}  // namespace msorb
