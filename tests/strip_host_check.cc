// The strip form of the FAST cell table (FrameGeom::strips, orb_host.cc) against the cell table it is derived from, on many
// geometries: every cell in exactly one strip, cells of a strip adjacent in one cell row, the strip's ROI and per-thread row
// count inside what fast_strip_kernel's LDS layout and flag words hold, the two LDS classes bounding their strips.
#include <cstdio>
#include <vector>

#include "../ms-slam_amd/csrc/orb_host.h"
using namespace msorb;

int main() {
    int bad = 0, with_strips = 0, n_geo = 0;
    const int dims[][2] = {{376, 1241}, {480, 752}, {400, 800}, {720, 1280}, {1080, 1920}, {360, 640}, {240, 320}, {333, 517},
                           {512, 512}, {600, 800}, {768, 1024}, {1200, 1600}, {480, 640}, {1024, 1280}, {2160, 3840}};
    for (const auto& d : dims)
        for (int nlev : {1, 4, 8})
            for (float sf : {1.2f, 1.5f, 2.0f}) {
                OrbParams P;
                P.init(2000, sf, nlev, 20, 7);
                FrameGeom g;
                if (!g.build(P, d[0], d[1])) continue;
                n_geo++;
                if (g.strips.empty()) continue;
                with_strips++;
                std::vector<int> owner(g.cells.size(), -1);
                for (size_t si = 0; si < g.strips.size(); si++) {
                    const StripDesc& t = g.strips[si];
                    const int cls = (int)si < g.strip_n_small ? 0 : 1;
                    if (t.ncell < 1 || t.ncell > kStripCells) bad++;
                    if (t.rh > g.strip_max_rh[cls]) bad++;
                    const int dh = t.rh - 6;
                    if (t.R < 1 || t.R > kStripMaxR || t.R * (kStripThreads / t.G) < dh || t.ncell * dh > kStripThreads) bad++;
                    if (4 * t.ndw + 8 > 4 * ((kStripCells * kStripMaxCellW + 6 + 3 + 3) / 4) + 8) bad++;
                    for (int c = 0; c < t.ncell; c++) {
                        const int id = t.cell0 + c;
                        if (id < 0 || id >= (int)g.cells.size() || owner[id] != -1) { bad++; continue; }
                        owner[id] = (int)si;
                        const CellDesc& cc = g.cells[id];
                        if (cc.level != t.level || cc.y0 != t.y0 || cc.rh != t.rh || cc.x0 != t.x0 + c * t.w_cell || cc.slot_off != t.slot_off[c]) bad++;
                        if (2 * (cc.rw - 6) * dh > g.strip_work_cap[cls]) bad++;
                        if (c == t.ncell - 1 && cc.x0 + cc.rw - t.x0 != t.rw) bad++;
                    }
                }
                for (int o : owner) bad += o < 0;
            }
    printf("geometries=%d with_strips=%d bad=%d\n", n_geo, with_strips, bad);
    return bad != 0;
}
