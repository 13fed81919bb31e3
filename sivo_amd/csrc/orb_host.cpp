// orb_host.cpp — the sequential part of ORB extraction that stays on the host:
// the quadtree ("OctTree") keypoint distribution of
// ORBextractor::DistributeOctTree / ExtractorNode::DivideNode
// (reference src/orbslam/ORBextractor.cc:488-750).  It is list surgery over a few
// thousand candidates per level ("stays on host").
//
// Data structure: an index arena.  A cell is an axis-aligned box [x0,x1) x [y0,y1)
// plus the (order-preserving) list of candidate indices inside it; cells live in one
// std::vector and are chained by prev/next indices in the order the reference's
// std::list would hold them (children are inserted at the front, the parent is
// unlinked), because that order defines both the split schedule and the output order.
//
// Tie rule: the reference sorts pair<size, node address>, i.e. equal sizes are ordered
// by heap address (not reproducible).  Here equal sizes are ordered by creation
// sequence (later-created = greater), the same documented rule the oracle uses.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "common.hpp"
#include "orb.hpp"

namespace sivo {
namespace {

// (Round 6: the candidates of a cell are a RANGE of one pooled index array — a split counts the four quadrants, takes four consecutive ranges
// at the end of the pool and scatters the parent's indices into them in order — instead of a std::vector per cell with its own heap block:
// thousands of small allocations per level were most of the quadtree's time, and the quadtree is half of a stand-alone extraction.  The
// arena and the pool are kept per thread between calls.)
struct Cell {
    int x0, y0, x1, y1;
    int kb = 0, ke = 0;      // candidates: pool[kb .. ke), indices into the candidate array, in arrival order
    bool leaf = false;       // holds exactly one candidate: never split again
    int prev = -1, next = -1;
    bool linked = false;
    int size() const { return ke - kb; }
};

struct Arena {
    std::vector<Cell> cells;
    std::vector<int> pool;
    int head = -1, tail = -1, count = 0;

    void reset(size_t n) {
        cells.clear(); pool.clear();
        head = tail = -1; count = 0;
        if (cells.capacity() < 4 * (n + 8)) cells.reserve(4 * (n + 8));
    }
    int make(int x0, int y0, int x1, int y1) {
        cells.emplace_back();
        Cell &c = cells.back();
        c.x0 = x0; c.y0 = y0; c.x1 = x1; c.y1 = y1;
        return (int)cells.size() - 1;   // the id doubles as the creation sequence number
    }
    void push_back(int id) {
        Cell &c = cells[id];
        c.prev = tail; c.next = -1; c.linked = true;
        if (tail >= 0) cells[tail].next = id; else head = id;
        tail = id; ++count;
    }
    void push_front(int id) {
        Cell &c = cells[id];
        c.next = head; c.prev = -1; c.linked = true;
        if (head >= 0) cells[head].prev = id; else tail = id;
        head = id; ++count;
    }
    int unlink(int id) {   // returns the successor
        Cell &c = cells[id];
        const int nx = c.next;
        if (c.prev >= 0) cells[c.prev].next = c.next; else head = c.next;
        if (c.next >= 0) cells[c.next].prev = c.prev; else tail = c.prev;
        c.linked = false; --count;
        return nx;
    }
};

// Split `id` into its four quadrants (NW, NE, SW, SE = the reference's n1..n4) and
// link the non-empty ones at the front in that order.  Returns through `grown` the
// children that hold more than one candidate.
void split(Arena &A, int id, const SivoKeyPoint *kp, std::vector<std::pair<int, int>> &grown, int *n_expand) {
    const int x0 = A.cells[id].x0, y0 = A.cells[id].y0, x1 = A.cells[id].x1, y1 = A.cells[id].y1;
    const int hx = (int)std::ceil((float)(x1 - x0) / 2), hy = (int)std::ceil((float)(y1 - y0) / 2);
    const int pb = A.cells[id].kb, pe = A.cells[id].ke;
    const int q[4] = {A.make(x0, y0, x0 + hx, y0 + hy), A.make(x0 + hx, y0, x1, y0 + hy),
                      A.make(x0, y0 + hy, x0 + hx, y1), A.make(x0 + hx, y0 + hy, x1, y1)};
    const float mx = (float)(x0 + hx), my = (float)(y0 + hy);
    int cnt[4] = {0, 0, 0, 0};
    for (int i = pb; i < pe; ++i) {
        const int k = A.pool[i];
        const bool west = kp[k].x < mx, north = kp[k].y < my;
        ++cnt[west ? (north ? 0 : 2) : (north ? 1 : 3)];
    }
    int at[4];
    const int base = (int)A.pool.size();
    A.pool.resize((size_t)base + (pe - pb));
    for (int i = 0, o = base; i < 4; ++i) { at[i] = o; A.cells[q[i]].kb = o; o += cnt[i]; A.cells[q[i]].ke = o; }
    for (int i = pb; i < pe; ++i) {
        const int k = A.pool[i];
        const bool west = kp[k].x < mx, north = kp[k].y < my;
        A.pool[at[west ? (north ? 0 : 2) : (north ? 1 : 3)]++] = k;
    }
    for (int i = 0; i < 4; ++i) {
        Cell &c = A.cells[q[i]];
        if (c.size() == 0) continue;
        if (c.size() == 1) c.leaf = true;
        A.push_front(q[i]);
        if (c.size() > 1) {
            if (n_expand) ++*n_expand;
            grown.emplace_back(c.size(), q[i]);
        }
    }
}

}  // namespace

int distribute_quadtree(const SivoKeyPoint *kp, int n, int min_x, int max_x, int min_y, int max_y, int target,
                        std::vector<SivoKeyPoint> &out) {
    out.clear();
    if (n <= 0) return 0;
    static thread_local Arena arena;
    Arena &A = arena;
    A.reset((size_t)n);
    const int n_ini = (int)std::round((float)(max_x - min_x) / (max_y - min_y));
    const float h_x = (float)(max_x - min_x) / n_ini;
    std::vector<int> roots(n_ini > 0 ? n_ini : 0);
    for (int i = 0; i < n_ini; ++i) {
        roots[i] = A.make((int)(h_x * (float)i), 0, (int)(h_x * (float)(i + 1)), max_y - min_y);
        A.push_back(roots[i]);
    }
    {   // the roots' candidates: count, then scatter in arrival order
        std::vector<int> cnt(roots.size(), 0), at(roots.size(), 0);
        for (int k = 0; k < n; ++k) ++cnt[(int)(kp[k].x / h_x)];
        A.pool.resize((size_t)n);
        for (size_t i = 0, o = 0; i < roots.size(); ++i) { at[i] = (int)o; A.cells[roots[i]].kb = (int)o; o += cnt[i]; A.cells[roots[i]].ke = (int)o; }
        for (int k = 0; k < n; ++k) A.pool[at[(int)(kp[k].x / h_x)]++] = k;
    }
    for (int it = A.head; it >= 0;) {
        Cell &c = A.cells[it];
        if (c.size() == 1) { c.leaf = true; it = c.next; }
        else if (c.size() == 0) it = A.unlink(it);
        else it = c.next;
    }

    std::vector<std::pair<int, int>> grown, prev;   // (size, cell id); id order == creation order
    bool done = false;
    while (!done) {
        int before = A.count, n_expand = 0;
        grown.clear();
        for (int it = A.head; it >= 0;) {
            if (A.cells[it].leaf) { it = A.cells[it].next; continue; }
            split(A, it, kp, grown, &n_expand);
            it = A.unlink(it);
        }
        if (A.count >= target || A.count == before) {
            done = true;
        } else if (A.count + n_expand * 3 > target) {
            // close to the target: split the most populated cells first, one at a time
            while (!done) {
                before = A.count;
                prev.swap(grown);
                grown.clear();
                std::sort(prev.begin(), prev.end());
                for (int j = (int)prev.size() - 1; j >= 0; --j) {
                    split(A, prev[j].second, kp, grown, nullptr);
                    A.unlink(prev[j].second);
                    if (A.count >= target) break;
                }
                if (A.count >= target || A.count == before) done = true;
            }
        }
    }
    // keep the strongest candidate of every cell (first one wins ties), in list order
    out.reserve(A.count);
    for (int it = A.head; it >= 0; it = A.cells[it].next) {
        const Cell &c = A.cells[it];
        int best = A.pool[c.kb];
        for (int k = c.kb + 1; k < c.ke; ++k)
            if (kp[A.pool[k]].response > kp[best].response) best = A.pool[k];
        out.push_back(kp[best]);
    }
    return (int)out.size();
}

}  // namespace sivo

extern "C" int sivo_orb_distribute(const SivoKeyPoint *keys, int n, int min_x, int max_x, int min_y, int max_y,
                                   int n_features, SivoKeyPoint *out, int capacity, int *n_out) {
    return sivo::guarded([&] {
        if (n < 0 || (n && !keys) || !n_out) throw std::invalid_argument("bad argument");
        if (max_x <= min_x || max_y <= min_y) throw std::invalid_argument("empty region");
        std::vector<SivoKeyPoint> res;
        sivo::distribute_quadtree(keys, n, min_x, max_x, min_y, max_y, n_features, res);
        *n_out = (int)res.size();
        if ((int)res.size() > capacity) return sivo::fail(SIVO_ERR_CAPACITY, "%zu keypoints, capacity %d", res.size(), capacity);
        std::copy(res.begin(), res.end(), out);
        return SIVO_OK;
    });
}
