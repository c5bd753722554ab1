// K3x: exact mode of the eikonal solve -- the reference's Fast Marching replayed literally, four units per wavefront
// (exact_march.h; reference CalSurfG.f90:288-487 `travel`, :587-759 `fouds2`, :768-921 the tree, :1287-1349 the hand-off).
// The engine runs it for the units whose fixed-point solve met an exact time tie (option exact_ties = 1) or for all of them
// (exact_ties = 2); it replaces the unit's refined snapshot (Tfin_r, S_r) and its compact coarse field with the march's own.
#include "kernels.h"

#include <algorithm>

#include "exact_march.h"
#include "receiver_core.h"

namespace dsa {

// ================================================================================================================================
// Round 4: FOUR units per wavefront.  Round 3's march -- one wavefront per unit -- was bound by the SCALAR unit (one per CU: ~1000 scalar
// instructions per accept, 1165 cycles per accept per CU measured against 1000 predicted), because everything sequential -- the tree, the
// statuses -- was made wave-uniform.  Here a unit owns a GROUP of sixteen lanes (the sixteen quadrant lanes of the stencil, as before) and
// a wavefront carries four groups: what was scalar is vector work that serves four units at once (the tree arithmetic is done alike by the
// sixteen lanes of a group, stores are issued by the group's first lane), and the four units' marches run in lockstep, each in its own
// exec-masked control flow.  Per accept step of a group:
//   * root from the tree (LDS), its coordinates, the record indices of its four neighbours and of the quadrant each lane owns;
//   * six loads per lane (the quadrant's four stencil records, the neighbour's slowness, its column's risti), in flight while
//   * the root leaves the tree (reference downtree, :800-858): one 16-byte read per level (both children); lane l of the group keeps the
//     move of level l, and all moves -- tree entry (LDS / global beyond lcap) and the moved node's status (its slot; reference nsts) -- are
//     stored in ONE pass behind the walk;
//   * the neighbours' statuses are read AFTER those stores (the wave's memory operations on one address keep their order), so a slot is
//     never stale from the root's removal -- the step log of the one-wavefront kernel is gone --, and their latency hides behind
//   * the quadrant candidates (exact_march.h: x_quad_candidates, fouds2's own expressions), the 4-lane minimum per neighbour (DPP);
//   * the four neighbours in the reference's order x-, x+, z-, z+: trial value stored, the node added (far) or moved up (in the tree;
//     reference updtree / addtree, towards the root while strictly smaller) -- WITHOUT a loop: the ancestors of a slot are known in advance,
//     the group reads fifteen of them at once, votes, and stores the moves in one pass (xg_sift_up); the sixteenth lane checks that the
//     node still sits where its status said (an earlier neighbour's moves of the same step may have pushed it down a level).
// Same tree, same insertion order, same comparisons: the same field as the reference's Fast Marching, bit for bit (tests/test_gpu_exact.py).
// The stages around the marches (resetting the fields, snapshot + hand-off, the compact copy) are kernels of their own, all lanes busy.

// exchange inside a group of four lanes: quad_perm [1,0,3,2] (the other k) and [2,3,0,1] (the other j)
__device__ __forceinline__ int x_dpp_other_k(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true); }
__device__ __forceinline__ int x_dpp_other_j(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true); }
__device__ __forceinline__ float x_dpp_min4(float v)
{
    float o = __int_as_float(x_dpp_other_k(__float_as_int(v)));
    v = (o < v) ? o : v;
    o = __int_as_float(x_dpp_other_j(__float_as_int(v)));
    return (o < v) ? o : v;
}

struct XStart { int id; float T; };          // one node of the coarse stage's starting tree (reference scan order ix outer, iz inner, :341-347)

// (explicit address spaces: a tree slot lives in LDS or in global memory, and with generic pointers the compiler folds the two reads into ONE
// flat load -- whose wait also covers every global load in flight, i.e. the stencil fetch the tree work is meant to hide)
#if defined(__HIP_DEVICE_COMPILE__)
#define DSA_GLB __attribute__((address_space(1)))
#else
#define DSA_GLB
#endif
struct XG {                                  // the march of one unit; every lane of its group holds the same values
    DSA_GLB XRec* F;                         // refined boxes: (T, status) records
    DSA_GLB unsigned* P;                     // propagation grid: ONE word per node (packed records, below)
    DSA_GLB const float* slow; DSA_GLB const float* risti;
    int nbz, nnx, nnz; unsigned nbz_inv;
    float ri, dnx, dnz;
    DSA_LDS XEntry* hl; DSA_GLB XEntry* hg;
    DSA_LDS XEntry* sc;                      // sixteen entries of LDS scratch: the levels of the tree beyond lcap, three at a time (xg_pop_root)
    int lcap, gcap;
    int lb;                                  // > 0: lcap = 2^lb - 1 and the global part of the tree is stored in BLOCKS (xg_gi); 0: slot by slot
    int ntr, err;
    unsigned pops;
    XEntry last;                             // tree[ntr], fetched at the end of the step before: the entry the next removal of the root sinks
    // pooled tiles (MD = 2; xg_tile_*): tile table, tile pool, the allocated tiles in allocation order (ring of tile numbers), the slots given back
    // (stack), tiles that must stay (bitmap: a receiver cell's corner lies in them); every lane of the group holds the same counters
    DSA_GLB unsigned short* tt; DSA_GLB unsigned* tp; DSA_GLB unsigned* ring; DSA_GLB unsigned short* freestk; DSA_GLB const unsigned* pins;
    int tcap, bump, nfree, rh, rt;           // slots of the pool; next never-used slot; stack size; ring head / tail (monotonic, index mod tcap)
    int nbx;
};

// Records of the propagation grid, round 4: one 32-bit word per node instead of (T, status) -- far 0xffffffff; alive the value's bits (a
// travel time: sign bit clear); in the tree 0x80000000 | slot.  A node in the tree needs no value of its own in the field: its trial value
// is its tree entry's key (fouds2 reads alive nodes only, and overwrites a trial value without looking at it, :758), and the node being
// accepted hands its value over in a register (XLane::rootj / rootk).  Half the bytes per unit: twice the units in flight where memory
// bounds them (4097^2: 67 MB per unit instead of 134), half the sectors behind a tile of neighbours.  PK = false: the refined boxes'
// (T, status) records, whose trial values the hand-off's snapshot wants (reference ttnr holds them, :1287).
constexpr unsigned kXFar = 0xffffffffu, kXInTree = 0x80000000u;
__device__ __forceinline__ XRec xg_unpack(unsigned w)
{
    XRec r;
    r.T = __uint_as_float(w);
    r.st = w == kXFar ? -1 : (int)w >= 0 ? 0 : (int)(w & 0xffffu);
    return r;
}
// MD: 0 = the refined boxes' (T, status) records, 1 = packed words over the whole propagation grid, 2 = packed words in POOLED TILES (round 5):
// the unit holds only the 8x8-node tiles its narrow band has touched and not yet left behind -- XG::tt maps a tile of the grid to a slot of
// the unit's tile pool XG::tp (kTNone: never touched, every node far; kTDone: every node alive, the slot given back) -- so that a times-only
// call on a large grid marches thousands of units where whole fields (67 MB each at 4097^2) let a few hundred march; see xg_tile_* below.
constexpr unsigned kTNone = 0xffffu, kTDone = 0xfffeu;
// word index of record id in the unit's tile pool; ~0u: the tile holds no slot
__device__ __forceinline__ unsigned xg_wa(const XG& m, int id)
{
    const unsigned t = m.tt[(unsigned)id >> 6];
    return t < kTDone ? (t << 6) | ((unsigned)id & 63u) : ~0u;
}
template <int MD> __device__ __forceinline__ unsigned xg_word(const XG& m, int id)
{
    if (MD == 2) {
        // (kTNone: every node far.  kTDone: every node alive -- the word's value is gone, and no stencil that is evaluated reads it: a node within
        // two steps of a node that is not alive lies in a tile that cannot have been given back; status "alive" is all the caller may use)
        const unsigned t = m.tt[(unsigned)id >> 6];
        return t < kTDone ? m.tp[(t << 6) | ((unsigned)id & 63u)] : t == kTDone ? 0u : kXFar;
    }
    return m.P[id];
}
template <int MD> __device__ __forceinline__ XRec xg_load(const XG& m, int id) { if (MD != 0) return xg_unpack(xg_word<MD>(m, id)); return m.F[id]; }
template <int MD> __device__ __forceinline__ int xg_status(const XG& m, int id) { if (MD != 0) return xg_unpack(xg_word<MD>(m, id)).st; return m.F[id].st; }
template <int MD> __device__ __forceinline__ void xg_store(XG& m, int id, unsigned w)
{
    if (MD == 2) { const unsigned a = xg_wa(m, id); if (a != ~0u) m.tp[a] = w; else m.err = 3; }      // (a store finds its tile allocated: xg_tile_need ran in front of it)
    else m.P[id] = w;
}
template <int MD> __device__ __forceinline__ void xg_set_alive(XG& m, int id, float T) { if (MD != 0) xg_store<MD>(m, id, __float_as_uint(T)); else m.F[id].st = 0; }
template <int MD> __device__ __forceinline__ void xg_set_slot(XG& m, int id, int s) { if (MD != 0) xg_store<MD>(m, id, kXInTree | (unsigned)s); else m.F[id].st = s; }
template <int MD> __device__ __forceinline__ void xg_set_trial(XG& m, int id, float T) { if (MD == 0) m.F[id].T = T; }

// Tree entries move as 8- / 16-byte vectors through pointers of an EXPLICIT address space.  (Copying an XEntry struct goes through its
// implicit copy constructor, i.e. through a generic reference: the compiler then folds "slot in LDS ? LDS read : global read" into ONE flat
// load of a selected generic pointer -- a load that counts on both memory counters, so that waiting for it also waits for every global
// load in flight: the stencil fetch the tree work is meant to hide.)
typedef float xf2 __attribute__((ext_vector_type(2)));
typedef float xf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ XEntry xg_entry(xf2 v) { return XEntry{ v.x, __float_as_int(v.y) }; }
__device__ __forceinline__ xf2 xg_vec(XEntry e) { xf2 v; v.x = e.key; v.y = __int_as_float(e.id); return v; }
// Where slot s > lcap of the tree lies in the unit's global part (in entries).  Round 5: with lcap = 2^lb - 1 (whole levels in LDS) the
// levels beyond are stored three at a time -- the fourteen descendants of a node p of level lb - 1 + 3 g (children, grandchildren, great-
// grandchildren) share one 128-byte line, in the order of the little tree the removal of the root walks (position = index in p's
// fifteen-node subtree - 2) -- so that a trip of the walk down, and the three nearest ancestors an update looks at, are one line instead of
// three: at full load the march moves 14 lines per accept, most of them tree entries (profiles/r05_pmc_march_full_load.txt).
// Lines of generation g follow those of the generations before: 2^(lb-1) (1 + 8 + ... + 8^(g-1)) of them.
__device__ __forceinline__ unsigned xg_line(const XG& m, unsigned p, int gq)
{
    return ((0x49249249u & ((1u << (3 * gq)) - 1u)) << (m.lb - 1)) + (p - (1u << (m.lb - 1 + 3 * gq)));
}
__device__ __forceinline__ unsigned xg_gi(const XG& m, int s)
{
    if (m.lb == 0) return (unsigned)(s - m.lcap - 1);
    const int t = (31 - __builtin_clz((unsigned)s)) - m.lb;          // level beyond the LDS part: 0, 1, 2, ...
    const int gq = (t * 11) >> 5;                                     // t / 3
    const int sh = t - 3 * gq + 1;
    const unsigned p = (unsigned)s >> sh;
    const unsigned li = ((unsigned)s & ((1u << sh) - 1u)) | (1u << sh);
    return (xg_line(m, p, gq) << 4) + li - 2u;
}
// slot s of the tree; the lanes of a group may ask for different slots.  The LDS read is unconditional (slot 1 stands in for a slot
// beyond the LDS part), the global one sits behind a branch the whole wavefront skips when no lane needs it
__device__ __forceinline__ XEntry xg_get(const XG& m, int s)
{
    xf2 v = *(DSA_LDS const xf2*)(m.hl + (s <= m.lcap ? s : 1));
    if (s > m.lcap) v = *(DSA_GLB const xf2*)(m.hg + xg_gi(m, s));
    return xg_entry(v);
}
// entry into slot s of the tree for the lanes with `on`; the lanes hold different (slot, entry) pairs.  LDS slot 0 is nobody's: lanes that
// have nothing for the LDS part write there instead of branching around the store
__device__ __forceinline__ void xg_put_tree(XG& m, bool on, int s, XEntry e)
{
    *(DSA_LDS xf2*)(m.hl + ((on && s <= m.lcap) ? s : 0)) = xg_vec(e);
    if (on && s > m.lcap) *(DSA_GLB xf2*)(m.hg + xg_gi(m, s)) = xg_vec(e);
}
// ... and the node's status (reference nsts: its slot)
template <int MD> __device__ __forceinline__ void xg_put_status(XG& m, bool on, int s, XEntry e)
{
    if (on) xg_set_slot<MD>(m, e.id, s);
}
template <int MD> __device__ __forceinline__ void xg_put(XG& m, bool on, int s, XEntry e)
{
    xg_put_tree(m, on, s, e);
    xg_put_status<MD>(m, on, s, e);
}
// the sixteen bits of a group in a wavefront-wide vote
__device__ __forceinline__ unsigned xg_vote16(bool c, int lane) { return (unsigned)(__ballot(c) >> (lane & 48)) & 0xffffu; }

// reference updtree / the tail of addtree (:768-790, :906-920): entry e, at slot s, moves towards the root while strictly smaller than
// its parent.  The ancestors of a slot are known in advance (s >> 1, s >> 2, ...), so the group reads fifteen of them at once (lane l:
// ancestor l + 1), votes on `e.key < ancestor's key`, and the run of ancestors BELOW the first one that says no -- the literal loop's
// moves; the reference's tree is not always a valid heap, so it is the first no that counts, not the last yes -- goes one level down in
// one store pass.  `check` (the entry was in the tree; s is the status read behind the root's removal): lane 15 reads slot s itself and
// checks that the node is still there -- an earlier neighbour's moves of this step may have pushed it down a level; then the status is
// read again (the wave's stores and loads on one address keep their order).  (A tree has at most sixteen levels: xg capacity.)
// the group's look at the path of slot s: lane l < 15 its ancestor l + 1, lane 15 (check) the slot itself
__device__ __forceinline__ XEntry xg_path(const XG& m, int s, bool check, int gl)
{
    const int a = gl < 15 ? (s >> (gl + 1)) : s;
    const bool have = a >= 1 && (gl < 15 || check);
    return xg_get(m, have ? a : 1);
}
// (one attempt with the path entries p: false when lane 15 found another node at slot s -- nothing is stored then; *moved: entries went down)
template <int MD> __device__ __forceinline__ bool xg_sift_apply(XG& m, XEntry e, int s, bool check, XEntry p, int gl, int lane, bool* moved)
{
    const int a = gl < 15 ? (s >> (gl + 1)) : s;
    const bool have = a >= 1 && (gl < 15 || check);
    const bool c = gl < 15 ? (have && e.key < p.key) : (check && p.id != e.id);
    const unsigned b = xg_vote16(c, lane);
    if (b & 0x8000u) return false;
    // one store pass: lane l < moves its ancestor's entry a level down (slot s >> l), lane `moves` the entry itself where it stops
    const int moves = __builtin_ctz(~b);
    XEntry w = p;
    if (gl == moves) w = e;
    xg_put<MD>(m, gl <= moves, s >> gl, w);
    *moved = moves > 0;
    return true;
}
template <int MD> __device__ __forceinline__ void xg_sift_up(XG& m, XEntry e, int s, bool check, int gl, int lane)
{
    bool moved;
    if (!xg_sift_apply<MD>(m, e, s, check, xg_path(m, s, check, gl), gl, lane, &moved)) {
        // (rare: the node was pushed down a level by an earlier neighbour of this step; its status says where to)
        s = xg_status<MD>(m, e.id);
        (void)xg_sift_apply<MD>(m, e, s, false, xg_path(m, s, false, gl), gl, lane, &moved);
    }
}
template <int MD> __device__ __forceinline__ void xg_add(XG& m, int id, float key, int gl, int lane)
{
    if (m.ntr + 1 > m.lcap + m.gcap) { m.err = 1; return; }
    m.ntr += 1;
    xg_sift_up<MD>(m, XEntry{ key, id }, m.ntr, false, gl, lane);
}
// reference downtree (:800-858): the last entry replaces the root and sinks; of two children with equal keys the left one is taken
// (`>`), a child moves up only when strictly smaller.  The walk down reads one 16-byte pair of children per level (lcap is odd: the
// children (tpc, tpc + 1), tpc even, lie on one side of the LDS / global split together, 16-byte aligned on either side) and leaves the
// move of level l with lane l of the group (at most fifteen moves and the sinking entry itself: sixteen levels); the moves -- tree
// entries and statuses -- are stored in one pass behind the walk.  Two loops, the levels in LDS and the levels beyond, so that the first
// carries no code of the second; no branches inside a level.
#define DSA_XG_LEVEL(PAIR, MORE)                                                             \
    {                                                                                        \
        const xf4 c = (PAIR);                                                                \
        const bool right = c.x > c.z;                                                        \
        const float ak = right ? c.z : c.x;                                                  \
        const int ai = __float_as_int(right ? c.w : c.y);                                    \
        tpc += right ? 1 : 0;                                                                \
        const bool mv = ak < e.key;                                                          \
        const bool cap = mv && gl == level;                                                  \
        mine.key = cap ? ak : mine.key; mine.id = cap ? ai : mine.id; mydst = cap ? tpp : mydst; myold = cap ? tpc : myold; \
        level += mv ? 1 : 0; tpp = mv ? tpc : tpp; tpc = mv ? 2 * tpc : m.ntr + 1;          \
        MORE                                                                                 \
    }
// What the walk leaves behind (XPop): lane l < moves holds the entry that went from slot `from` up to slot `to` = from / 2 at level l, lane
// `moves` the sunk entry and its slot; the TREE is updated (LDS part and the part beyond), the statuses of the moved nodes are the caller's
// to store (xg_put_status) -- at the end of the step, behind every load of it: on this hardware a load's wait also waits for the stores
// issued before it, and a scattered store into a field that no cache holds takes as long as a miss.
struct XPop { XEntry mine; int to, from, moves; };
__device__ __forceinline__ XPop xg_pop_root(XG& m, int gl)
{
    XPop P;
    P.mine = m.last; P.to = 0; P.from = 0; P.moves = -1;
    if (m.ntr == 1) { m.ntr = 0; return P; }
    const XEntry e = m.last;
    m.ntr -= 1;
    int tpp = 1, tpc = 2, level = 0, mydst = 0, myold = 0;
    XEntry mine = e;
    const int lim = m.ntr < m.lcap ? m.ntr : m.lcap;
    // two levels per LDS round trip while the four grandchildren exist and lie in LDS: the pair of children and both pairs under them are
    // read together, the second level picks its pair by the first level's choice
    while (2 * tpc + 3 <= lim) {
        const xf4 c = *(DSA_LDS const xf4*)(m.hl + tpc), g0 = *(DSA_LDS const xf4*)(m.hl + 2 * tpc), g1 = *(DSA_LDS const xf4*)(m.hl + 2 * tpc + 2);
        const bool r1 = c.x > c.z;
        const float k1 = r1 ? c.z : c.x;
        const int i1 = __float_as_int(r1 ? c.w : c.y), c1 = tpc + (r1 ? 1 : 0);
        const bool mv1 = k1 < e.key;
        const xf4 gp = r1 ? g1 : g0;
        const bool r2 = gp.x > gp.z;
        const float k2 = r2 ? gp.z : gp.x;
        const int i2 = __float_as_int(r2 ? gp.w : gp.y), c2 = 2 * c1 + (r2 ? 1 : 0);
        const bool mv2 = mv1 && k2 < e.key;
        const bool cap1 = mv1 && gl == level, cap2 = mv2 && gl == level + 1;
        mine.key = cap1 ? k1 : cap2 ? k2 : mine.key; mine.id = cap1 ? i1 : cap2 ? i2 : mine.id;
        mydst = cap1 ? tpp : cap2 ? c1 : mydst; myold = cap1 ? c1 : cap2 ? c2 : myold;
        level += (mv1 ? 1 : 0) + (mv2 ? 1 : 0);
        tpp = mv2 ? c2 : mv1 ? c1 : tpp;
        tpc = mv2 ? 2 * c2 : m.ntr + 1;
    }
    while (tpc < lim) DSA_XG_LEVEL(*(DSA_LDS const xf4*)(m.hl + tpc), )
    if (tpc == m.ntr && tpc <= m.lcap) {                      // (an only child, in LDS)
        const XEntry a = xg_entry(*(DSA_LDS const xf2*)(m.hl + tpc));
        const bool mv = a.key < e.key, cap = mv && gl == level;
        mine.key = cap ? a.key : mine.key; mine.id = cap ? a.id : mine.id; mydst = cap ? tpp : mydst; myold = cap ? tpc : myold;
        level += mv ? 1 : 0; tpp = mv ? tpc : tpp;
        tpc = m.ntr + 1;
    }
    // the levels beyond the LDS part, three per memory round trip: fourteen lanes fetch the descendants of tpp down to its great-
    // grandchildren (a slot beyond the tree's end reads as +inf: an only child then wins its "pair" as the reference's tail rule has it)
    // into LDS scratch laid out as a little tree of its own (children at 2..3, their children at 4..7, 8..15), and the walk goes on there
    while (tpc <= m.ntr) {
        const int d = gl < 2 ? 1 : gl < 6 ? 2 : 3, off = gl - ((1 << d) - 2);
        const int slot = (tpp << d) + off;
        xf2 v; v.x = kInf; v.y = 0.0f;
        // (blocked: tpp stands at level lb - 1 + 3 g -- the walk enters this loop from the last LDS level and goes on three levels at a time --
        // and its fourteen descendants are the first fourteen entries of its line)
        unsigned gi = (unsigned)(slot - m.lcap - 1);
        if (m.lb) { const int lp = 31 - __builtin_clz((unsigned)tpp); gi = (xg_line(m, (unsigned)tpp, ((lp - m.lb + 1) * 11) >> 5) << 4) + (unsigned)gl; }
        if (gl < 14 && slot <= m.ntr) v = *(DSA_GLB const xf2*)(m.hg + gi);
        *(DSA_LDS xf2*)(m.sc + (gl < 14 ? (1 << d) + off : gl - 14)) = v;
        int lpc = 2;
        while (lpc < 16 && tpc <= m.ntr) DSA_XG_LEVEL(*(DSA_LDS const xf4*)(m.sc + lpc), lpc = 2 * (lpc + (right ? 1 : 0));)
    }
    if (gl == level) { mine = e; mydst = tpp; }
    xg_put_tree(m, gl <= level, mydst, mine);
    P.mine = mine; P.to = mydst; P.from = myold; P.moves = level;
    return P;
}
#undef DSA_XG_LEVEL

// value of the lane at byte address 4 * lane of the wavefront
__device__ __forceinline__ int xg_from_lane(int byte_addr, int v) { return __builtin_amdgcn_ds_bpermute(byte_addr, v); }

// ---- pooled tiles (MD = 2) ------------------------------------------------------------------------------------------------------------------------
// A tile can go when no stencil will read it again: fouds2 reads a node's neighbours up to two steps away in x or z, so the alive nodes of tile T
// are read only by nodes of T and of its four edge neighbours -- when T and those four are alive to the last node (tiles beyond the grid count as
// alive, nodes of a tile beyond the grid's edge do not count at all), T's slot goes back; a tile that holds a receiver cell's corner stays
// (k_xreceivers reads it at the end).  Nothing is counted while marching: when the unit needs a slot and has none, it looks at the OLDEST tiles
// of its ring -- sixteen lanes read a tile's 64 words in one load -- and frees those that qualify; the others go to the ring's tail.
// tile `t` alive to the last node that lies inside the grid?  (allocated tiles: their words; kTDone: yes; kTNone: no; beyond the grid: yes)
__device__ __forceinline__ bool xg_tile_all_alive(const XG& m, int tx, int tz, int gl, int lane)
{
    if (tx < 0 || tz < 0 || tx >= m.nbx || tz >= m.nbz) return true;
    const unsigned t = m.tt[tx * m.nbz + tz];
    if (t == kTDone) return true;
    if (t == kTNone) return false;
    const uint4 w = *(DSA_GLB const uint4*)(m.tp + ((size_t)t << 6) + 4 * gl);          // lane gl: records 4 gl .. 4 gl + 3 (row gl / 2 of the tile, four z)
    const int ix = (tx << 3) + (gl >> 1), iz0 = (tz << 3) + 4 * (gl & 1);
    const bool inx = ix < m.nnx;
    bool ok = true;
    ok = ok && (!(inx && iz0 + 0 < m.nnz) || (int)w.x >= 0);
    ok = ok && (!(inx && iz0 + 1 < m.nnz) || (int)w.y >= 0);
    ok = ok && (!(inx && iz0 + 2 < m.nnz) || (int)w.z >= 0);
    ok = ok && (!(inx && iz0 + 3 < m.nnz) || (int)w.w >= 0);
    return xg_vote16(!ok, lane) == 0u;
}
// look at up to `look` of the oldest tiles; returns with freed slots on the stack (or none: the caller reports the pool as too small)
__device__ __forceinline__ void xg_tile_collect(XG& m, int look, int gl, int lane)
{
    const bool lead = gl == 0;
    const int n = m.rt - m.rh < look ? m.rt - m.rh : look;
    for (int k = 0; k < n; ++k) {
        const unsigned tile = m.ring[(unsigned)m.rh % (unsigned)m.tcap];
        m.rh += 1;
        const int tx = (int)(tile / (unsigned)m.nbz), tz = (int)(tile - (unsigned)tx * (unsigned)m.nbz);
        const bool pinned = (m.pins[tile >> 5] >> (tile & 31u)) & 1u;
        bool go = !pinned && xg_tile_all_alive(m, tx, tz, gl, lane);
        go = go && xg_tile_all_alive(m, tx - 1, tz, gl, lane) && xg_tile_all_alive(m, tx + 1, tz, gl, lane);
        go = go && xg_tile_all_alive(m, tx, tz - 1, gl, lane) && xg_tile_all_alive(m, tx, tz + 1, gl, lane);
        if (go) {
            const unsigned slot = m.tt[tile];
            if (lead) { m.tt[tile] = (unsigned short)kTDone; m.freestk[m.nfree] = (unsigned short)slot; }
            m.nfree += 1;
        } else {
            if (lead) m.ring[(unsigned)m.rt % (unsigned)m.tcap] = tile;
            m.rt += 1;
        }
    }
}
// make sure record id's tile has a slot (the group's lanes all pass the same id); the new tile starts far
__device__ __forceinline__ void xg_tile_need(XG& m, int id, int gl, int lane)
{
    const unsigned tile = (unsigned)id >> 6;
    if (m.tt[tile] < kTDone) return;
    if (m.nfree == 0 && m.bump >= m.tcap) {
        // no slot left: the oldest tiles, a few dozen at a time, until one goes (or the whole ring has been looked at twice)
        for (int tries = 0; m.nfree == 0 && tries < 2 * m.tcap; tries += 32) xg_tile_collect(m, 32, gl, lane);
        if (m.nfree == 0) { m.err = 2; return; }
    }
    unsigned slot;
    if (m.nfree > 0) { m.nfree -= 1; slot = m.freestk[m.nfree]; }
    else { slot = (unsigned)m.bump; m.bump += 1; }
    const uint4 far4 = { kXFar, kXFar, kXFar, kXFar };
    *(DSA_GLB uint4*)(m.tp + ((size_t)slot << 6) + 4 * gl) = far4;
    if (gl == 0) { m.tt[tile] = (unsigned short)slot; m.ring[(unsigned)m.rt % (unsigned)m.tcap] = tile; }
    m.rt += 1;
}

// what a lane is within its group, fixed for the whole march: lane 4 q + 2 j + k owns quadrant (j: the x- / x+ side, k: the z- / z+ side)
// of neighbour q (x-, x+, z-, z+) of whatever node is being accepted
struct XLane {
    int gl, j, k;
    int dx, dz;          // neighbour q's offset from the accepted node
    int xs, zs;          // the quadrant's direction: -1 / +1
    bool rootj, rootk;   // the accepted node IS this quadrant's x / z stencil neighbour (it lies on the far side of neighbour q)
    int base;            // byte address of the group's first lane (ds_bpermute)
};
__device__ __forceinline__ XLane xg_lane(int lane)
{
    XLane L;
    L.gl = lane & 15;
    const int q = L.gl >> 2;
    L.j = (L.gl >> 1) & 1; L.k = L.gl & 1;
    L.dx = q == 0 ? -1 : q == 1 ? 1 : 0;
    L.dz = q == 2 ? -1 : q == 3 ? 1 : 0;
    L.xs = L.j ? 1 : -1; L.zs = L.k ? 1 : -1;
    L.rootj = (q == 0 && L.j == 1) || (q == 1 && L.j == 0);
    L.rootk = (q == 2 && L.k == 1) || (q == 3 && L.k == 0);
    L.base = (lane & 48) << 2;
    return L;
}
// record index of node (iz0, ix0), 0-based (eikonal_core.h rec_index; the tile product on the 24-bit multiplier)
__device__ __forceinline__ int xg_rec(int nbz, int iz0, int ix0)
{
    return (int)(((__umul24((unsigned)ix0 >> 3, (unsigned)nbz) + ((unsigned)iz0 >> 3)) << 6) | (((unsigned)ix0 & 7u) << 3) | ((unsigned)iz0 & 7u));
}

// One accept step of reference travel (:417-485) for the group's unit.  No branches outside the tree work: a lane's loads are issued
// whatever the grid's edges say (a node outside reads the root's record, which exists, and the value is dropped)
#ifdef DSA_X_CLOCKS
#define DSA_XCLK(k) { const unsigned long long now_ = __builtin_readcyclecounter(); xc[k] += now_ - xt; xt = now_; }
#else
#define DSA_XCLK(k)
#endif
template <int MD> __device__ __forceinline__ void xg_accept_root(XG& m, XEntry root, int iz0, int ix0, const XLane& L, int lane
#ifdef DSA_X_CLOCKS
                                               , unsigned long long* xc, unsigned long long& xt
#endif
                                               )
{
    const bool lead = L.gl == 0;
    // my neighbour, my quadrant of its stencil: coordinates, who lies inside the grid, record indices
    const int mx0 = ix0 + L.dx, mz0 = iz0 + L.dz;
    const bool in = (unsigned)mx0 < (unsigned)m.nnx && (unsigned)mz0 < (unsigned)m.nnz;
    const int xj = mx0 + L.xs, xj2 = xj + L.xs, zk = mz0 + L.zs, zk2 = zk + L.zs;
    const bool inj = in && (unsigned)xj < (unsigned)m.nnx, inj2 = in && (unsigned)xj2 < (unsigned)m.nnx;
    const bool ink = in && (unsigned)zk < (unsigned)m.nnz, ink2 = in && (unsigned)zk2 < (unsigned)m.nnz;
    const int mid = in ? xg_rec(m.nbz, mz0, mx0) : root.id;
    const int idj = inj ? xg_rec(m.nbz, mz0, xj) : root.id, idj2 = inj2 ? xg_rec(m.nbz, mz0, xj2) : root.id;
    const int idk = ink ? xg_rec(m.nbz, zk, mx0) : root.id, idk2 = ink2 ? xg_rec(m.nbz, zk2, mx0) : root.id;
    // seven loads per lane in flight while the root leaves the tree: the quadrant's stencil, the neighbour's slowness and status
    XRec vj = xg_load<MD>(m, idj), vk = xg_load<MD>(m, idk);
    const XRec vj2 = xg_load<MD>(m, idj2), vk2 = xg_load<MD>(m, idk2);
    unsigned tile_mid = 0u;                    // (pooled tiles: the slot of my neighbour's tile, or kTNone: then its tile has to be allocated before it enters the tree)
    int st_pre;
    if (MD == 2) {
        tile_mid = m.tt[(unsigned)mid >> 6];
        st_pre = xg_unpack(tile_mid < kTDone ? m.tp[(tile_mid << 6) | ((unsigned)mid & 63u)] : tile_mid == kTDone ? 0u : kXFar).st;
    } else st_pre = xg_status<MD>(m, mid);
    if (MD != 0) {     // (packed records: the node being accepted is still "in the tree" in the field; its value is the root's key)
        vj.T = L.rootj ? root.key : vj.T;
        vk.T = L.rootk ? root.key : vk.T;
    }
    // (the tree's last entry but one: the entry the NEXT removal sinks, unless this step adds nodes or writes that slot -- see the step's end)
    const XEntry spare = xg_get(m, m.ntr > 1 ? m.ntr - 1 : 1);
    const float slown = m.slow[mid], risti = m.risti[in ? mx0 : 0];
    DSA_XCLK(1)
    const int ntr_old = m.ntr;
    // (pooled tiles: the word addresses the step's stores will need are looked up here, in front of the first store -- a look-up behind a store would
    // wait for it, and a scattered store into memory no cache holds takes as long as a miss; the root's own word first)
    unsigned wa_root = 0u, wa_move = 0u;
    if (MD == 2) wa_root = xg_wa(m, root.id);
    const XPop P = xg_pop_root(m, L.gl);
    if (MD == 2) wa_move = xg_wa(m, L.gl <= P.moves ? P.mine.id : root.id);
    DSA_XCLK(2)
    // my neighbour's slot after the root's removal: its status was fetched BEFORE it, so if the neighbour is one of the entries the walk
    // moved -- from a slot on the walk's path to that slot's parent, or (the tree's last entry) to where it sank -- the slot follows it.
    // The path's slot at the depth of mine sits with the lane of that level.
    int st = in ? st_pre : 0;
    {
        const int lv = 30 - __builtin_clz(st > 1 ? st : 2);                       // level of the move that would have taken slot st: depth - 1
        const int sunk = xg_from_lane(L.base + 4 * (P.moves < 0 ? 0 : P.moves), P.to);
        const int from = xg_from_lane(L.base + 4 * (lv & 15), P.from);
        st = (st > 1 && lv < P.moves && from == st) ? (st >> 1) : st;
        st = (st_pre == ntr_old && in && P.moves >= 0) ? sunk : st;
    }
#ifdef DSA_X_CLOCKS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DSA_XCLK(3)
#endif
    if (m.ntr + 4 > m.lcap + m.gcap) { m.err = 1; return; }
    // The four neighbours, each in its OWN four lanes (lane r of neighbour q's quad): its slot in the tree -- added (far: the next free
    // slots, in the reference's order x-, x+, z-, z+) or already in it --, and the part of its path an update normally touches: lanes 0..2 the
    // three nearest ancestors, lane 3 (a node already in the tree) the slot itself.  One LDS read, one memory round trip where the paths
    // leave the LDS part, issued before the stencil arithmetic so that it hides behind it.
    const int r4 = L.gl & 3, q4 = L.gl & 12;
    const bool isnew = st < 0;
    const unsigned newq = xg_vote16(isnew && r4 == 0, lane);
    const int sq = st == 0 ? 0 : isnew ? m.ntr + 1 + __popc(newq & ((1u << q4) - 1u)) : st;
    const int aq = r4 < 3 ? (sq >> (r4 + 1)) : (isnew ? 0 : sq);
    const XEntry pq = xg_get(m, aq >= 1 ? aq : 1);
    unsigned wa_pq = 0u;
    if (MD == 2) wa_pq = xg_wa(m, aq >= 1 ? pq.id : root.id);
    XQuadState s;
    s.ej = inj; s.ek = ink;
    s.aj = inj && (vj.st == 0 || L.rootj);  s.oj = inj2 && vj2.st == 0;
    s.ak = ink && (vk.st == 0 || L.rootk);  s.ok = ink2 && vk2.st == 0;
    s.tj = vj.T; s.tj2 = vj2.T; s.tk = vk.T; s.tk2 = vk2.T;
    const int dk = (s.ek && !s.ak) ? 1 : 0, dj = (s.ej && !s.aj) ? 1 : 0;
    // (bitwise or: with `||` the exchange would sit behind a branch, and a lane that skips it reads as 0 to its partner)
    const int dk_other = x_dpp_other_k(dk), dj_other = x_dpp_other_j(dj);
    const bool k_dead = (dk | dk_other) != 0, j_dead = (dj | dj_other) != 0;
    const NodeGeom g = { m.ri, risti, m.dnx, m.dnz };
    const float c = x_quad_lane(s, k_dead, j_dead, L.j, L.k, slown, g);
    const float trial = x_dpp_min4(in ? c : kInf);
    DSA_XCLK(4)
    // The updates of the four neighbours at once.  Each quad votes on `trial < ancestor's key` (lane 3: is the node still where its status
    // said) and finds how many levels its node climbs: 0..2 -- or it asks for the sequential way (3: more ancestors to look at; a node not
    // at its slot).  The reference updates the neighbours one after the other, each seeing the tree the one before left: the quads may go
    // together only if no EARLIER neighbour writes a slot a LATER one reads or writes.  An update of node at slot s that climbs `up` levels
    // writes s, s/2 .. s >> up and reads one ancestor more; two such chains meet at the depth of the slots' lowest common ancestor, so six
    // lanes -- one per pair of neighbours -- compare that depth with the depths the two chains reach.
    const bool cq = r4 < 3 ? (aq >= 1 && trial < pq.key) : (aq >= 1 && pq.id != mid);
    const unsigned vq = (xg_vote16(cq, lane) >> q4) & 15u;
    const int up = __builtin_ctz(~(vq & 7u));
    bool seq = st != 0 && (up == 3 || (vq & 8u) != 0u);
    {
        // pair p of (earlier, later) neighbours: (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
        const int pe = L.gl < 3 ? 0 : L.gl < 5 ? 1 : 2, pl = L.gl < 3 ? L.gl + 1 : L.gl < 5 ? L.gl - 1 : 3;
        const int se = xg_from_lane(L.base + 16 * pe, sq), ue = xg_from_lane(L.base + 16 * pe, up);
        const int sl = xg_from_lane(L.base + 16 * pl, sq), ul = xg_from_lane(L.base + 16 * pl, up);
        const int de = 31 - __builtin_clz(se | 1), dl = 31 - __builtin_clz(sl | 1), dm = de < dl ? de : dl;
        const unsigned x = (unsigned)(se >> (de - dm)) ^ (unsigned)(sl >> (dl - dm));
        const int lca = dm - (x ? 32 - __builtin_clz(x) : 0);
        const int reach_e = de - ue, reach_l = dl - ul - 1;
        seq = seq || (L.gl < 6 && se > 0 && sl > 0 && lca >= (reach_e > reach_l ? reach_e : reach_l));
    }
    const bool sequential = xg_vote16(seq, lane) != 0u;
    if (MD == 2) {
        // (pooled tiles) a neighbour that enters the tree may lie in a tile the unit has not touched yet: one tile in ~64 steps
        const unsigned needq = xg_vote16(isnew && r4 == 0 && tile_mid >= kTDone, lane);
        if (needq) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idq = xg_from_lane(L.base + 16 * q, mid);
                if ((needq >> (4 * q)) & 1u) xg_tile_need(m, idq, L.gl, lane);
            }
            if (m.err) return;
            tile_mid = m.tt[(unsigned)mid >> 6];
        }
    }
    // From here on the step only stores (but for the rare sequential way): the root alive, the statuses of the entries its removal moved,
    // the neighbours' trial values (fouds2 overwrites them unconditionally, :758: the first lane of each neighbour's four stores it) ...
    // (the paths fetched above are waited for HERE, before the first store: a wait further down would also cover the stores)
    __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0)
    if (MD == 2) {
        if (lead) m.tp[wa_root] = __float_as_uint(root.key);
        if (L.gl <= P.moves) m.tp[wa_move] = kXInTree | (unsigned)P.to;
    } else {
        if (lead) xg_set_alive<MD>(m, root.id, root.key);
        xg_put_status<MD>(m, L.gl <= P.moves, P.to, P.mine);
    }
    if (r4 == 0 && st != 0) xg_set_trial<MD>(m, mid, trial);
    if (!sequential) {
        // ... and the neighbours' updates, one store pass for the four: lane r < up its ancestor's entry a level down (slot s >> r), lane `up`
        // the node itself where it stops
        XEntry w = pq;
        if (r4 == up) w = XEntry{ trial, mid };
        const bool wr = st != 0 && r4 <= up;
        if (MD == 2) {
            xg_put_tree(m, wr, sq >> r4, w);
            const unsigned wa_w = r4 == up ? ((tile_mid << 6) | ((unsigned)mid & 63u)) : wa_pq;
            if (wr) m.tp[wa_w] = kXInTree | (unsigned)(sq >> r4);
        } else xg_put<MD>(m, wr, sq >> r4, w);
        m.ntr += __popc(newq);
        // the tree's last entry for the next step, without a load behind these stores: it is what this step wrote there (a new node's chain
        // ends at the last slot; else an update's or the removal's move may have), or else the entry fetched at the step's start
        const bool h1 = wr && (sq >> r4) == m.ntr, h2 = L.gl <= P.moves && P.to == m.ntr;
        const unsigned v1 = xg_vote16(h1, lane), v2 = xg_vote16(h2, lane);
        const int src = L.base + 4 * (__builtin_ctz((v1 ? v1 : v2) | 0x10000u) & 15);
        const int lk = xg_from_lane(src, __float_as_int(h1 ? w.key : P.mine.key)), li = xg_from_lane(src, h1 ? w.id : P.mine.id);
        m.last = (v1 | v2) ? XEntry{ __int_as_float(lk), li } : spare;
    } else {
        // the sequential way (reference order, each neighbour reading its path after the one before has stored): lane 4 q of the group
        // holds neighbour q's values
#ifdef DSA_X_CLOCKS
        xc[7] += 1;
#endif
        const int tb = __float_as_int(trial);
        int stq = xg_from_lane(L.base, st), st_b = xg_from_lane(L.base + 16, st), st_c = xg_from_lane(L.base + 32, st), st_d = xg_from_lane(L.base + 48, st);
        int idq = xg_from_lane(L.base, mid), id_b = xg_from_lane(L.base + 16, mid), id_c = xg_from_lane(L.base + 32, mid), id_d = xg_from_lane(L.base + 48, mid);
        float trq = __int_as_float(xg_from_lane(L.base, tb)), tr_b = __int_as_float(xg_from_lane(L.base + 16, tb));
        float tr_c = __int_as_float(xg_from_lane(L.base + 32, tb)), tr_d = __int_as_float(xg_from_lane(L.base + 48, tb));
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {
            if (stq != 0) {
                const bool fresh = stq < 0;
                m.ntr += fresh ? 1 : 0;
                xg_sift_up<MD>(m, XEntry{ trq, idq }, fresh ? m.ntr : xg_status<MD>(m, idq), !fresh, L.gl, lane);
            }
            stq = st_b; st_b = st_c; st_c = st_d; st_d = 0;
            idq = id_b; id_b = id_c; id_c = id_d;
            trq = tr_b; tr_b = tr_c; tr_c = tr_d;
        }
        if (m.ntr > 0) m.last = xg_get(m, m.ntr);
    }
    DSA_XCLK(5)
    m.pops += 1u;
}

// the marches of up to four units per wavefront, until every tree is empty.  REFINED: travel(urg = 1) on the refined boxes from the four
// corners of the source cell; the reference's exit -- the root lies on an edge of the box that is not an edge of the model by the literal
// test of :396-407 -- marks that node alive and stops.  Otherwise: travel(urg = 2) on the propagation grid from the hand-off's tree
// the pooled tiles' arrays of a batch (MD = 2): unit slot j at tt + j * tt_stride etc.
struct XTilePool { unsigned short* tt; size_t tt_stride; unsigned* tp; unsigned* ring; unsigned short* freestk; unsigned* pins; size_t pins_stride; int tcap; };

template <bool REFINED, bool POOLED = false>
__global__ __launch_bounds__(64) void k_xmarch(GridDesc g, BatchPtrs b, const int* __restrict__ units, int n, const float* __restrict__ slow_all,
                                               size_t field_stride, const float* __restrict__ risti_c, unsigned* pool, size_t pool_stride,
                                               XEntry* heap_pool, int gcap, int lcap, const XStart* __restrict__ starts, const int* __restrict__ nstart,
                                               int32_t* xinfo, unsigned long long* clk, XTilePool tpool, int gstride, int lb, const int* n_dev)
{
    extern __shared__ unsigned char x_lds[];
    if (n_dev) { const int c = *n_dev; n = c < n ? c : n; }      // (the hand-off's replay list: its length is known to the device only)
    if (n <= 0) return;
    const int lane = threadIdx.x, grp = lane >> 4;
    const bool lead = (lane & 15) == 0;
    const int slot = blockIdx.x * 4 + grp;
    const bool live = slot < n;
    const int s = units[live ? slot : n - 1];
    XG m;
    m.hl = (DSA_LDS XEntry*)x_lds + (size_t)grp * (size_t)(lcap + 1);
    m.sc = (DSA_LDS XEntry*)x_lds + (size_t)4 * (size_t)(lcap + 1) + (size_t)grp * 16;
    m.last = XEntry{ 0.0f, 0 };
    m.lcap = lcap; m.hg = (DSA_GLB XEntry*)(heap_pool + (size_t)(live ? slot : 0) * (size_t)gstride); m.gcap = gcap; m.lb = lb;
    m.ntr = 0; m.err = 0; m.pops = 0u; m.ri = g.earth;
    m.F = nullptr; m.P = nullptr;
    m.tt = nullptr; m.tp = nullptr; m.ring = nullptr; m.freestk = nullptr; m.pins = nullptr; m.tcap = 0; m.bump = 0; m.nfree = 0; m.rh = 0; m.rt = 0; m.nbx = g.nbx;
    constexpr int MD = REFINED ? 0 : POOLED ? 2 : 1;
    int rnx = 0, rnz = 0, oxl = 0, oxh = 0, ozl = 0, ozh = 0;
    if (REFINED) {
        const SourceDesc* sd = b.src + s;
        m.F = (DSA_GLB XRec*)(b.F_r + (size_t)s * kRefRecs); m.slow = (DSA_GLB const float*)(b.slow_r + (size_t)s * kRefRecs); m.risti = (DSA_GLB const float*)(b.risti_r + (size_t)s * kRefMax);
        rnx = sd->rnx; rnz = sd->rnz; oxl = sd->open_xlo; oxh = sd->open_xhi; ozl = sd->open_zlo; ozh = sd->open_zhi;
        x_set_grid(m, sd->nbz_r, rnx, rnz); m.dnx = sd->rdnx; m.dnz = sd->rdnz;
        if (live) {
            // the four corners of the source cell, values with distances in radians (:360-375)
            const float* vc = b.vcorner + (size_t)s * 4;
            float vss[2][2];
            for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) vss[i][j] = vc[i * 2 + j];
            const float dsx = sd->dsx_r, dsz = sd->dsz_r;
            const float vsrc = bilinear4(vss, m.dnx, m.dnz, dsx, dsz);
            const int isx = sd->isx_r, isz = sd->isz_r;
            for (int i = 1; i <= 2; ++i)
                for (int j = 1; j <= 2; ++j) {
                    const float ds = sqrtf(sq(dsx - (float)(i - 1) * m.dnx) + sq(dsz - (float)(j - 1) * m.dnz));
                    const float t = 2.0f * ds / (vss[i - 1][j - 1] + vsrc);
                    const int id = rec_index(m.nbz, isz - 2 + j, isx - 2 + i);
                    if (lead) m.F[id].T = t;
                    xg_add<MD>(m, id, t, lane & 15, lane);
                }
        }
    } else {
        const SourceDesc* sd = b.src + s;
        if (POOLED) {
            const size_t j = live ? (size_t)slot : 0;
            m.tt = (DSA_GLB unsigned short*)(tpool.tt + j * tpool.tt_stride); m.tp = (DSA_GLB unsigned*)(tpool.tp + j * ((size_t)tpool.tcap << 6));
            m.ring = (DSA_GLB unsigned*)(tpool.ring + j * (size_t)tpool.tcap); m.freestk = (DSA_GLB unsigned short*)(tpool.freestk + j * (size_t)tpool.tcap);
            m.pins = (DSA_GLB const unsigned*)(tpool.pins + j * tpool.pins_stride); m.tcap = tpool.tcap;
        } else m.P = (DSA_GLB unsigned*)(pool + (size_t)(live ? slot : 0) * pool_stride);
        m.slow = (DSA_GLB const float*)(slow_all + (size_t)sd->period * field_stride); m.risti = (DSA_GLB const float*)risti_c;
        x_set_grid(m, g.nbz, g.nnx, g.nnz); m.dnx = g.dnx; m.dnz = g.dnz;
        if (live) {
            const int cnt = nstart[slot];
            const XStart* st = starts + (size_t)slot * kXStage;
            for (int q = 0; q < cnt; ++q) {
                const XStart e = st[q];
                if (POOLED) {
                    // (the hand-off lists every node it puts on the propagation grid, the alive ones with ~id: their tiles get slots here)
                    const int id = e.id < 0 ? ~e.id : e.id;
                    xg_tile_need(m, id, lane & 15, lane);
                    if (m.err) break;
                    if (e.id < 0) { if (lead) xg_set_alive<MD>(m, id, e.T); continue; }
                }
                xg_add<MD>(m, e.id, e.T, lane & 15, lane);
            }
        }
    }
    const XLane L = xg_lane(lane);
    if (live && m.ntr > 0) m.last = xg_get(m, m.ntr);
#ifdef DSA_X_CLOCKS
    unsigned long long xc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, xt = __builtin_readcyclecounter();
#endif
    bool active = live;
    for (;;) {
        active = active && m.ntr > 0 && m.err == 0;
        if (!__any(active)) break;
        if (active) {
            const XEntry root = xg_entry(*(DSA_LDS const xf2*)(m.hl + 1));
            int iz0, ix0;
            x_coords(m, root.id, &iz0, &ix0);
            bool stop = false;
            if (REFINED) {
                const int iz = iz0 + 1, ix = ix0 + 1;
                stop = (ix == 1 && oxl) || (ix == rnx && oxh) || (iz == 1 && ozl) || (iz == rnz && ozh);
            }
            if (stop) { if (lead) xg_set_alive<MD>(m, root.id, root.key); active = false; }
#ifdef DSA_X_CLOCKS
            else { DSA_XCLK(0) xg_accept_root<MD>(m, root, iz0, ix0, L, lane, xc, xt); }
#else
            else xg_accept_root<MD>(m, root, iz0, ix0, L, lane);
#endif
        }
    }
#ifdef DSA_X_CLOCKS
    if (!REFINED && clk && lane == 0 && live) for (int q = 0; q < 8; ++q) clk[(size_t)s * kClockSlots + 8 + q] = xc[q];
#endif
    if (live && lead) {
        if (REFINED) { xinfo[4 * s + 0] = (int)m.pops; xinfo[4 * s + 2] = m.err; xinfo[4 * s + 3] = 0; }
        else { xinfo[4 * s + 1] = (int)m.pops; if (m.err) xinfo[4 * s + 2] = m.err; }
    }
}

// every record far, value 0 (the reference's nsts = -1): the refined boxes of the batch's units and their pool slots (pooled tiles: the
// tile tables -- no tile touched -- and the bitmaps of the tiles that must stay, filled by k_xpins)
__global__ __launch_bounds__(256) void k_xfill(BatchPtrs b, const int* __restrict__ units, unsigned* pool, size_t pool_stride, int nrec, XTilePool tpool, const int* n_dev)
{
    const int slot = blockIdx.y;
    if (n_dev && slot >= *n_dev) return;
    const int s = units[slot];
    const uint4 v = { 0u, 0xffffffffu, 0u, 0xffffffffu };              // two records {0.0f, -1}
    const uint4 far4 = { kXFar, kXFar, kXFar, kXFar };                 // four packed records (or eight tile-table entries kTNone)
    uint4* const Fr = (uint4*)(b.F_r + (size_t)s * kRefRecs);
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = t; i < (size_t)kRefRecs / 2; i += step) Fr[i] = v;
    if (tpool.tt) {
        uint4* const tt4 = (uint4*)(tpool.tt + (size_t)slot * tpool.tt_stride);
        for (size_t i = t; i < tpool.tt_stride / 8; i += step) tt4[i] = far4;
        unsigned* const pins = tpool.pins + (size_t)slot * tpool.pins_stride;
        for (size_t i = t; i < tpool.pins_stride; i += step) pins[i] = 0u;
        return;
    }
    if (!pool) return;                 // (the refined boxes alone: launch_refined_replay)
    uint4* const Fc = (uint4*)(pool + (size_t)slot * pool_stride);
    for (size_t i = t; i < (size_t)nrec / 4; i += step) Fc[i] = far4;
}

// pooled tiles: the tiles a unit's receivers will read at the end (the four corners of every receiver's cell, receiver_core.h) must keep their slots
__global__ __launch_bounds__(64) void k_xpins(GridDesc g, BatchPtrs b, const int* __restrict__ units, const RayDesc* __restrict__ rays, XTilePool tpool)
{
    const int slot = blockIdx.x;
    const SourceDesc sd = b.src[units[slot]];
    unsigned* const pins = tpool.pins + (size_t)slot * tpool.pins_stride;
    for (int r = threadIdx.x; r < sd.nrec; r += 64) {
        const RayDesc rd = rays[sd.first_ray + r];
        if (!(rd.flags & kRayTime)) continue;
        int irx = (int)((rd.rx - g.gox) / g.dnx) + 1, irz = (int)((rd.rz - g.goz) / g.dnz) + 1;
        if (irx < 1 || irx > g.nnx || irz < 1 || irz > g.nnz) continue;
        if (irx == g.nnx) irx -= 1;
        if (irz == g.nnz) irz -= 1;
        for (int k = 0; k < 2; ++k)
            for (int l = 0; l < 2; ++l) {
                const unsigned tile = (unsigned)rec_index(g.nbz, irz - 1 + l, irx - 1 + k) >> 6;
                atomicOr(&pins[tile >> 5], 1u << (tile & 31u));
            }
    }
}

// between the two marches, one workgroup per unit: the snapshot the ray tracer reads (reference ttnr / nstsr, :1287-1288), every sgdl-th
// refined node -- status and, for status >= 0, value -- onto the propagation grid (:1293-1303), alive nodes that touch a far node back
// into the narrow band (:1332-1349), and the starting tree's nodes in the reference's scan order (:341-347)
__global__ __launch_bounds__(64) void k_xhandoff(GridDesc g, BatchPtrs b, const int* __restrict__ units, unsigned* pool, size_t pool_stride,
                                                 XStart* starts, int* nstart, int pooled)
{
    __shared__ int stage_st[kXStage];
    __shared__ float stage_T[kXStage];
    const int slot = blockIdx.x, lane = threadIdx.x;
    const int s = units[slot];
    const SourceDesc sd = b.src[s];
    const size_t rr = (size_t)kRefMax * kRefMax;
    const XRec* const Fr = (const XRec*)(b.F_r + (size_t)s * kRefRecs);
    float* const Tfin = b.Tfin_r + (size_t)s * rr;
    int8_t* const Sr = b.S_r + (size_t)s * rr;
    const int nref = sd.rnx * sd.rnz;
    for (int id = lane; id < nref; id += 64) {
        const int ix = id / sd.rnz, iz = id - ix * sd.rnz;
        const XRec r = Fr[rec_index(sd.nbz_r, iz, ix)];
        Sr[id] = (int8_t)(r.st < 0 ? -1 : r.st == 0 ? 0 : 1);
        Tfin[id] = r.st >= 0 ? r.T : kInf;
    }
    const int bxn = (sd.rnx - 1) / kSgdl + 1, bzn = (sd.rnz - 1) / kSgdl + 1;
    for (int q = lane; q < bxn * bzn; q += 64) {
        const int l = (q / bzn) * kSgdl, k = (q - (q / bzn) * bzn) * kSgdl;      // 0-based refined node
        const XRec r = Fr[rec_index(sd.nbz_r, k, l)];
        stage_st[q] = r.st < 0 ? -1 : r.st == 0 ? 0 : 1;
        stage_T[q] = r.T;
    }
    __syncthreads();
    unsigned promote = 0u;                                                      // bit t: this lane's t-th node
    for (int q = lane, t = 0; q < bxn * bzn; q += 64, ++t) {
        if (stage_st[q] != 0) continue;
        const int bx = q / bzn, bz = q - bx * bzn;
        const int cx = sd.vnl + bx, cz = sd.vnt + bz;                           // 1-based node of the propagation grid
        const int dx[4] = { -1, 1, 0, 0 }, dz[4] = { 0, 0, -1, 1 };
        for (int d = 0; d < 4; ++d) {
            const int nx = cx + dx[d], nz = cz + dz[d];
            if (nx < 1 || nx > g.nnx || nz < 1 || nz > g.nnz) continue;
            const int ox = bx + dx[d], oz = bz + dz[d];
            const bool inbox = ox >= 0 && ox < bxn && oz >= 0 && oz < bzn;
            if (!inbox || stage_st[ox * bzn + oz] == -1) promote |= 1u << t;
        }
    }
    __syncthreads();
    for (int q = lane, t = 0; q < bxn * bzn; q += 64, ++t) if ((promote >> t) & 1u) stage_st[q] = 1;
    __syncthreads();
    unsigned* const Fc = pool + (size_t)slot * pool_stride;
    XStart* const out = starts + (size_t)slot * kXStage;
    int count = 0;
    for (int q0 = 0; q0 < bxn * bzn; q0 += 64) {
        const int q = q0 + lane;
        const bool have = q < bxn * bzn;
        const int st = have ? stage_st[q] : -1;
        const int bx = have ? q / bzn : 0, bz = have ? q - bx * bzn : 0;
        const int id = rec_index(g.nbz, sd.vnt + bz - 1, sd.vnl + bx - 1);
        const float tq = have ? stage_T[q] : 0.0f;
        if (pooled) {
            // (pooled tiles: the march gives the tiles their slots, so the alive nodes go on the list too, marked ~id, in the same scan order)
            const unsigned long long mask = __ballot(st >= 0);
            if (st >= 0) out[count + __popcll(mask & ((1ull << lane) - 1ull))] = XStart{ st == 0 ? ~id : id, tq };
            count += __popcll(mask);
            continue;
        }
        if (st == 0) Fc[id] = __float_as_uint(tq);
        const unsigned long long mask = __ballot(st > 0);
        if (st > 0) out[count + __popcll(mask & ((1ull << lane) - 1ull))] = XStart{ id, tq };      // (its word in the field: set when the march adds it to the tree)
        count += __popcll(mask);
    }
    if (lane == 0) nstart[slot] = count;
}

// the unit's compact coarse field: plain values, no exceptional nodes (every node was accepted once, in order)
__global__ __launch_bounds__(256) void k_xfinish(GridDesc g, BatchPtrs b, const int* __restrict__ units, const unsigned* pool, size_t pool_stride, int nrec)
{
    const int slot = blockIdx.y;
    const int s = units[slot];
    const unsigned* const Fc = pool + (size_t)slot * pool_stride;
    float* const T_c = b.T_c + (size_t)s * nrec;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)nrec; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned w = Fc[i];
        T_c[i] = (int)w >= 0 ? __uint_as_float(w) : kInf;
    }
}

// the batch's receiver times straight from the marched fields (reference srtimes, receiver_core.h): a call that wants nothing but times
// needs no compact copy of its units' fields -- at 4097^2 that copy is 67 MB per unit
struct MarchFieldT {
    const unsigned* F;
    __device__ __forceinline__ float operator()(int id) const { const unsigned w = F[id]; return (int)w >= 0 ? __uint_as_float(w) : kInf; }
};
struct MarchTilesT {
    const unsigned short* tt; const unsigned* tp;
    __device__ __forceinline__ float operator()(int id) const
    {
        const unsigned t = tt[(unsigned)id >> 6];
        if (t >= kTDone) return kInf;                  // (cannot happen for a receiver's corner: k_xpins keeps those tiles)
        const unsigned w = tp[((size_t)t << 6) | ((unsigned)id & 63u)];
        return (int)w >= 0 ? __uint_as_float(w) : kInf;
    }
};
__global__ __launch_bounds__(64) void k_xreceivers(GridDesc g, BatchPtrs b, const int* __restrict__ units, const unsigned* pool, size_t pool_stride,
                                                   const RayDesc* __restrict__ rays, const float* __restrict__ veln_all, size_t veln_stride, float dpl,
                                                   float* __restrict__ out, int32_t* __restrict__ err, XTilePool tpool)
{
    const int slot = blockIdx.x;
    const int s = units[slot];
    const SourceDesc sd = b.src[s];
    const MarchFieldT field{ pool + (size_t)slot * pool_stride };
    const MarchTilesT tiles{ tpool.tt + (size_t)slot * tpool.tt_stride, tpool.tp + (size_t)slot * ((size_t)tpool.tcap << 6) };
    const float* veln = veln_all + (size_t)sd.period * veln_stride;
    for (int r = threadIdx.x; r < sd.nrec; r += 64) {
        const RayDesc rd = rays[sd.first_ray + r];
        if (!(rd.flags & kRayTime)) continue;
        float t;
        const bool ok = tpool.tt ? receiver_time_f(g, sd.scx, sd.scz, rd, tiles, veln, dpl, &t) : receiver_time_f(g, sd.scx, sd.scz, rd, field, veln, dpl, &t);
        if (!ok) { atomicExch(err, sd.first_ray + r + 1); t = 0.0f; }
        out[rd.data] = t;
    }
}

// entries of a unit's global tree part in the blocked layout (xg_gi): whole lines up to the one that holds slot lcap + gcap
size_t exact_heap_blocked_entries(int lcap, int gcap)
{
    int lb = 0;
    while ((1 << lb) < lcap + 1) ++lb;
    const unsigned smax = (unsigned)(lcap + gcap);
    int lev = 31;
    while (!((smax >> lev) & 1u)) --lev;
    const int t = lev - lb, gq = t < 0 ? 0 : t / 3;
    // every line of the generations before the last one's, and of that one up to the parent of slot smax at the generation's deepest level
    size_t lines = ((size_t)(0x49249249u & ((1u << (3 * gq)) - 1u)) << (lb - 1));
    const unsigned plev = (unsigned)(lb - 1 + 3 * gq);
    const unsigned pmax = t < 0 ? (1u << plev) : std::min<unsigned>((2u << plev) - 1u, std::max<unsigned>(smax >> (t - 3 * gq + 1), 1u << plev));
    // (a shallower row of the same generation reaches further to the right than the deepest one: take the whole generation when it has more than one row)
    lines += (t - 3 * gq > 0 ? (size_t)1 << plev : (size_t)(pmax - (1u << plev)) + 1);
    return lines * 16;
}
size_t exact_lds_bytes(int lcap) { return (size_t)4 * (size_t)(lcap + 1 + 16) * sizeof(XEntry); }
size_t exact_start_bytes() { return (size_t)kXStage * sizeof(XStart); }
// pooled tiles, per marching unit: the tile table (two bytes per tile of the grid, a multiple of sixteen bytes), tcap tiles of 256 bytes, the ring
// (four bytes per slot), the stack of free slots (two), the bitmap of the tiles that stay
size_t exact_tile_table_entries(int ntile) { return ((size_t)ntile + 7) & ~(size_t)7; }
size_t exact_tile_unit_bytes(int ntile, int tcap) { return exact_tile_table_entries(ntile) * 2 + (size_t)tcap * (256 + 4 + 2) + (((size_t)ntile + 31) / 32) * 4; }

void launch_exact(const GridDesc& g, const BatchPtrs& b, const int* d_units, int n, const float* d_slow_all, size_t field_stride,
                  const float* d_risti_c, void* d_pool, size_t pool_stride, void* d_heap_pool, int gcap, int lcap, void* d_starts, int* d_nstart,
                  int32_t* d_xinfo, unsigned long long* d_clocks, const XReceivers* rc, bool compact_copy, hipStream_t stream, const XTiles* tiles,
                  int gstride, int lb)
{
    if (n <= 0) return;
    if (gstride <= 0) { gstride = gcap; lb = 0; }
    const size_t lds = exact_lds_bytes(lcap);
    if (lds > 48 * 1024) {   // (per device: set every time)
        (void)hipFuncSetAttribute((const void*)k_xmarch<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)k_xmarch<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)k_xmarch<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const int nrec = g.nbx * g.nbz * kTileRecs;
    const int ntile = g.nbx * g.nbz;
    XTilePool tp{};
    if (tiles) {
        tp.tt = (unsigned short*)tiles->tt; tp.tt_stride = exact_tile_table_entries(ntile); tp.tp = (unsigned*)tiles->tp; tp.ring = (unsigned*)tiles->ring;
        tp.freestk = (unsigned short*)tiles->freestk; tp.pins = (unsigned*)tiles->pins; tp.pins_stride = ((size_t)ntile + 31) / 32; tp.tcap = tiles->tcap;
    }
    const int fill_blocks = (int)std::min<size_t>(((size_t)nrec / 2 + 255) / 256, 64);
    hipLaunchKernelGGL(k_xfill, dim3(fill_blocks, n), dim3(256), 0, stream, b, d_units, (unsigned*)d_pool, pool_stride, nrec, tp, (const int*)nullptr);
    if (tiles && rc) hipLaunchKernelGGL(k_xpins, dim3(n), dim3(64), 0, stream, g, b, d_units, rc->rays, tp);
    const int waves = (n + 3) / 4;
    hipLaunchKernelGGL((k_xmarch<true, false>), dim3(waves), dim3(64), lds, stream, g, b, d_units, n, d_slow_all, field_stride, d_risti_c, (unsigned*)d_pool, pool_stride,
                       (XEntry*)d_heap_pool, gcap, lcap, (const XStart*)d_starts, (const int*)d_nstart, d_xinfo, d_clocks, tp, gstride, lb, (const int*)nullptr);
    hipLaunchKernelGGL(k_xhandoff, dim3(n), dim3(64), 0, stream, g, b, d_units, (unsigned*)d_pool, pool_stride, (XStart*)d_starts, d_nstart, tiles ? 1 : 0);
    if (tiles)
        hipLaunchKernelGGL((k_xmarch<false, true>), dim3(waves), dim3(64), lds, stream, g, b, d_units, n, d_slow_all, field_stride, d_risti_c, (unsigned*)d_pool, pool_stride,
                           (XEntry*)d_heap_pool, gcap, lcap, (const XStart*)d_starts, (const int*)d_nstart, d_xinfo, d_clocks, tp, gstride, lb, (const int*)nullptr);
    else
        hipLaunchKernelGGL((k_xmarch<false, false>), dim3(waves), dim3(64), lds, stream, g, b, d_units, n, d_slow_all, field_stride, d_risti_c, (unsigned*)d_pool, pool_stride,
                           (XEntry*)d_heap_pool, gcap, lcap, (const XStart*)d_starts, (const int*)d_nstart, d_xinfo, d_clocks, tp, gstride, lb, (const int*)nullptr);
    if (compact_copy && !tiles) hipLaunchKernelGGL(k_xfinish, dim3(fill_blocks, n), dim3(256), 0, stream, g, b, d_units, (const unsigned*)d_pool, pool_stride, nrec);
    if (rc) hipLaunchKernelGGL(k_xreceivers, dim3(n), dim3(64), 0, stream, g, b, d_units, (const unsigned*)d_pool, pool_stride, rc->rays, rc->veln_all, rc->veln_stride,
                               rc->dpl, rc->out, rc->err, tp);
}

// (round 6, last) the refined boxes of the units the hand-off's probe listed (d_list[0] = how many, known to the device only; d_list[1 ..] = the units),
// by the march: the records b.F_r of those units hold the march's (T, status) afterwards (kernels.h launch_handoff; stage_kernels.hip k_handoff_replay
// turns them into a refined stage that ended by itself).  A unit's tree: 1023 slots in LDS, `gcap` more at d_heap + slot * gcap.
void launch_refined_replay(const GridDesc& g, const BatchPtrs& b, const int32_t* d_list, int cap, void* d_heap, int gcap, int32_t* d_xinfo, hipStream_t stream)
{
    if (cap <= 0) return;
    const int lcap = 1023;
    const size_t lds = exact_lds_bytes(lcap);
    XTilePool tp{};
    hipLaunchKernelGGL(k_xfill, dim3(64, cap), dim3(256), 0, stream, b, (const int*)(d_list + 1), (unsigned*)nullptr, (size_t)0, 0, tp, (const int*)d_list);
    hipLaunchKernelGGL((k_xmarch<true, false>), dim3((cap + 3) / 4), dim3(64), lds, stream, g, b, (const int*)(d_list + 1), cap, (const float*)nullptr, (size_t)0, (const float*)nullptr,
                       (unsigned*)nullptr, (size_t)0, (XEntry*)d_heap, gcap, lcap, (const XStart*)nullptr, (const int*)nullptr, d_xinfo, (unsigned long long*)nullptr, tp, gcap, 0, (const int*)d_list);
}

}  // namespace dsa
