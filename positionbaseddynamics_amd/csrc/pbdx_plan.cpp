// pbdx_plan.cpp -- planner of the colour-fused tile schedule (see pbdx_plan.h).  Host only.
#include "pbdx_plan.h"
#include "pbdx_internal.h"
#include "../../include/pbdx_debug.h"
#include <algorithm>
#include <memory>
#include <atomic>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>

namespace pbdx {
namespace {

struct Graph
{
	uint32_t n = 0;
	uint32_t nc = 0;
	uint32_t ncolours = 0;
	const std::vector<PlanBatch> *batches = nullptr;
	std::vector<uint32_t> batch_base;     // nb + 1
	std::vector<uint16_t> cbatch;         // constraint id -> batch
	std::vector<uint32_t> adj_off, adj;   // particle -> constraint ids, ascending (= colour order)

	inline const PlanBatch &batch_of(uint32_t cid) const { return (*batches)[cbatch[cid]]; }
	inline uint32_t colour(uint32_t cid) const { return batch_of(cid).colour; }
	inline uint32_t bodies(uint32_t cid, const uint32_t *&out) const
	{
		const uint32_t b = cbatch[cid];
		const PlanBatch &pb = (*batches)[b];
		const uint32_t nbod = type_info(pb.type)->num_bodies;
		out = pb.idx + (size_t)(cid - batch_base[b]) * nbod;
		return nbod;
	}
};

// ---- bank-aware slot order of one step (pbdx_plan.h, LDS bank model) ---------------------------------------
// `cnt` independent constraints with `nb` endpoints each (h[c * nb + j] = tile-local slot of endpoint j) and optionally a table-record key per
// constraint (tkey: records with different keys that fall into the same class tcls conflict; 0xffffffff = none).  Returns in `order` a permutation of
// 0 .. cnt-1: lane l of the step's l / 64-th wave-pass projects constraint order[l].  Greedy: lanes are filled in turn, every lane takes the first of the
// next kWindow unassigned constraints (original order) whose endpoints collide with nothing its read group (16 lanes, banks = slot mod 16) and its write
// group (8 lanes, slot mod 8) already hold, or the one with the fewest collisions.  Never affects results (a colour's constraints are independent).
// Bank-aware NUMBERING of a run of LDS slots: `ids` (ascending particle ids) are to occupy slots start, start + 1, ...; they are permuted so that as
// many slots as possible satisfy slot mod 16 == id mod 16.  The bank class of a particle then follows its id across the regions of a tile (interior,
// boundary, halo), so that on a mesh with regular ids (grids: neighbours at constant id offsets) the endpoints of every constraint of one shape sit at
// constant CLASS offsets everywhere in the tile -- which is what lets BankOrder below form conflict-free groups.  Ids of a run that is contiguous stay in
// ascending order (a rotation by less than 16 at most): the fill's HBM reads remain as coalesced as before.  Any numbering is valid; never affects results.
inline void class_order(uint32_t *ids, uint32_t count, uint32_t start, std::vector<uint32_t> &tmp)
{
	if (count < 2) return;
	uint32_t n[16] = {}, head[16], off[17];
	for (uint32_t i = 0; i < count; i++) n[ids[i] & 15u]++;
	off[0] = 0;
	for (uint32_t q = 0; q < 16; q++) { off[q + 1] = off[q] + n[q]; head[q] = off[q]; }
	tmp.resize(count);
	{
		uint32_t fill[16];
		for (uint32_t q = 0; q < 16; q++) fill[q] = off[q];
		for (uint32_t i = 0; i < count; i++) tmp[fill[ids[i] & 15u]++] = ids[i];      // (stable: ascending inside a class)
	}
	for (uint32_t sl = 0; sl < count; sl++)
	{
		uint32_t q = (start + sl) & 15u;
		if (head[q] == off[q + 1])
		{
			// the wanted class has run out: the class with most particles left gives one up
			uint32_t best = 0, left = 0;
			for (uint32_t k = 0; k < 16; k++) if (off[k + 1] - head[k] > left) { left = off[k + 1] - head[k]; best = k; }
			q = best;
		}
		ids[sl] = tmp[head[q]++];
	}
}

struct BankOrder
{
	// Unassigned constraints by SHAPE (the bank classes of the other endpoints relative to the first: constraints of one shape whose first endpoints
	// fall into different classes collide in no endpoint) and by the class of the first endpoint; original order inside a queue.
	struct Shape { uint32_t sig, count, left; std::vector<uint32_t> q[16]; uint32_t head[16]; };
	std::vector<Shape> shapes;
	std::vector<uint32_t> sig_of, rank;
	void run(uint32_t cnt, uint32_t nb, const uint32_t *h, const uint32_t *tkey, const uint32_t *tcls, std::vector<uint32_t> &order)
	{
		order.resize(cnt);
		// shapes, most frequent first
		sig_of.resize(cnt);
		for (uint32_t i = 0; i < cnt; i++)
		{
			uint32_t sg = 0;
			for (uint32_t j = 1; j < nb; j++) sg = sg * 16u + ((h[(size_t)i * nb + j] - h[(size_t)i * nb]) & 15u);
			sig_of[i] = sg;
		}
		rank.assign(sig_of.begin(), sig_of.end());
		std::sort(rank.begin(), rank.end());
		size_t used = 0;
		for (size_t i = 0; i < rank.size();)
		{
			size_t e = i;
			while (e < rank.size() && rank[e] == rank[i]) e++;
			if (used == shapes.size()) shapes.emplace_back();
			Shape &sh = shapes[used++];
			sh.sig = rank[i]; sh.count = sh.left = (uint32_t)(e - i);
			for (uint32_t q = 0; q < 16; q++) { sh.q[q].clear(); sh.head[q] = 0; }
			i = e;
		}
		std::sort(shapes.begin(), shapes.begin() + used, [](const Shape &x, const Shape &y) { return x.count != y.count ? x.count > y.count : x.sig < y.sig; });
		// (signature -> shape: binary search over a sorted copy would do; the shape count is small, a linear map by signature is built instead)
		rank.assign(65536u >> (nb == 2 ? 12 : nb == 3 ? 8 : 4), 0xffffffffu);
		for (size_t k = 0; k < used; k++) rank[shapes[k].sig] = (uint32_t)k;
		for (uint32_t i = 0; i < cnt; i++) shapes[rank[sig_of[i]]].q[h[(size_t)i * nb] & 15u].push_back(i);
		uint16_t rmask[4][4]; uint8_t wmask[8][4];
		uint32_t tk[4][16];                // per read group and class: the record key that holds it (0xffffffff = free)
		auto cost_of = [&](uint32_t c, uint32_t rg, uint32_t wg)
		{
			uint32_t cost = 0;
			for (uint32_t j = 0; j < nb; j++)
			{
				const uint32_t hh = h[(size_t)c * nb + j];
				cost += (rmask[rg][j] >> (hh & 15u)) & 1u;
				cost += (wmask[wg][j] >> (hh & 7u)) & 1u;
			}
			if (tkey && tkey[c] != 0xffffffffu) { const uint32_t o = tk[rg][tcls[c] & 15u]; cost += (o != 0xffffffffu && o != tkey[c]) ? 1u : 0u; }
			return cost;
		};
		size_t cur = 0;                    // first shape with constraints left
		for (uint32_t l = 0; l < cnt; l++)
		{
			const uint32_t lane = l & 63u;
			if (lane == 0u) { memset(rmask, 0, sizeof(rmask)); memset(wmask, 0, sizeof(wmask)); memset(tk, 0xff, sizeof(tk)); }
			const uint32_t rg = lds_read_group(lane), wg = lds_write_group(lane);
			while (cur < used && !shapes[cur].left) cur++;
			// candidates: shape by shape from the current one; inside a shape the class lane mod 16 first (lanes of a group then differ by construction)
			size_t bs = 0; uint32_t bq = 0, bk = 0, best_cost = 0xffffffffu, evaluated = 0;
			for (size_t si = cur; si < used && best_cost && evaluated < kBudget; si++)
			{
				Shape &sh = shapes[si];
				if (!sh.left) continue;
				for (uint32_t dq = 0; dq < 16 && best_cost; dq++)
				{
					const uint32_t q = (lane + dq) & 15u;
					const uint32_t avail = (uint32_t)sh.q[q].size() - sh.head[q];
					const uint32_t look = std::min<uint32_t>(avail, dq == 0 ? 2u : 1u);
					for (uint32_t k = 0; k < look; k++)
					{
						const uint32_t cst = cost_of(sh.q[q][sh.head[q] + k], rg, wg);
						evaluated++;
						if (cst < best_cost) { best_cost = cst; bs = si; bq = q; bk = k; if (!cst) break; }
					}
				}
			}
			Shape &sh = shapes[bs];
			std::vector<uint32_t> &qq = sh.q[bq];
			const uint32_t c = qq[sh.head[bq] + bk];
			for (uint32_t k = bk; k > 0; k--) qq[sh.head[bq] + k] = qq[sh.head[bq] + k - 1];      // (keeps the queue in original order)
			sh.head[bq]++; sh.left--;
			order[l] = c;
			for (uint32_t j = 0; j < nb; j++)
			{
				const uint32_t hh = h[(size_t)c * nb + j];
				rmask[rg][j] |= (uint16_t)(1u << (hh & 15u));
				wmask[wg][j] |= (uint8_t)(1u << (hh & 7u));
			}
			if (tkey && tkey[c] != 0xffffffffu && tk[rg][tcls[c] & 15u] == 0xffffffffu) tk[rg][tcls[c] & 15u] = tkey[c];
		}
	}
	static constexpr uint32_t kBudget = 96;
};

// The constraints a closure has already collected: an open-addressing set stamped with the closure's serial number (no reset between closures).  Until
// round 6 this was a stamp word per constraint of the whole schedule and per thread -- 24 MB per thread at 6 M constraints, first touched (page faults)
// at the start of every build, for closures that hold some ten thousand constraints.
struct CidSet
{
	std::vector<uint32_t> key, ser;
	uint32_t mask = 0, used = 0, serial_of_used = 0;
	void init(uint32_t cap_log2) { key.assign((size_t)1 << cap_log2, 0u); ser.assign((size_t)1 << cap_log2, 0u); mask = (1u << cap_log2) - 1u; used = 0; serial_of_used = 0; }
	void clear_all() { std::fill(ser.begin(), ser.end(), 0u); used = 0; }
	// true if `cid` was not in the set of closure `serial` yet (and is now)
	inline bool insert(uint32_t cid, uint32_t serial)
	{
		if (serial_of_used != serial) { serial_of_used = serial; used = 0; }
		uint32_t h = (cid * 2654435761u) & mask;
		while (ser[h] == serial)
		{
			if (key[h] == cid) return false;
			h = (h + 1u) & mask;
		}
		key[h] = cid; ser[h] = serial;
		if (++used * 2u > mask) grow(serial);
		return true;
	}
	void grow(uint32_t serial)
	{
		std::vector<uint32_t> ok, os;
		ok.swap(key); os.swap(ser);
		const uint32_t nm = mask * 2u + 1u;
		key.assign((size_t)nm + 1u, 0u); ser.assign((size_t)nm + 1u, 0u); mask = nm;
		for (size_t i = 0; i < ok.size(); i++)
			if (os[i] == serial)
			{
				uint32_t h = (ok[i] * 2654435761u) & mask;
				while (ser[h] == serial) h = (h + 1u) & mask;
				key[h] = ok[i]; ser[h] = serial;
			}
	}
};

struct Scratch
{
	std::vector<uint32_t> stamp_p, local_of;
	CidSet cset;
	BankOrder bank; std::vector<uint32_t> bank_h, bank_key, bank_cls, bank_order;      // bank-aware slot order of the step being emitted
	uint32_t serial = 0;
	std::vector<std::vector<uint32_t>> bucket;
	std::vector<uint32_t> halo;

	void init(const Graph &g)
	{
		stamp_p.assign(g.n, 0);
		cset.init(17);
		local_of.assign(g.n, 0);
		serial = 0;
	}
};

// Backward dependency closure of the owned particles over colours [c_lo, c1).  `after(c)` is called
// when colour c is final: bucket[c - c_lo .. c1 - c_lo) hold the constraints (sorted by id) and
// `halo` the non-owned particles of the closure of [c, c1).
template <class F>
void closure(const Graph &g, Scratch &s, const uint32_t *owned, uint32_t n_owned, uint32_t c_lo, uint32_t c1, F &&after)
{
	if (++s.serial == 0)
	{
		std::fill(s.stamp_p.begin(), s.stamp_p.end(), 0u);
		s.cset.clear_all();
		s.serial = 1;
	}
	const uint32_t serial = s.serial;
	if (s.bucket.size() < c1 - c_lo) s.bucket.resize(c1 - c_lo);
	for (uint32_t c = 0; c < c1 - c_lo; c++) s.bucket[c].clear();
	s.halo.clear();
	auto add = [&](uint32_t p, uint32_t c_limit)
	{
		s.stamp_p[p] = serial;
		for (uint32_t a = g.adj_off[p]; a < g.adj_off[p + 1]; a++)
		{
			const uint32_t cid = g.adj[a];
			const uint32_t col = g.colour(cid);
			if (col >= c_limit) break;
			if (col < c_lo) continue;
			if (s.cset.insert(cid, serial)) s.bucket[col - c_lo].push_back(cid);
		}
	};
	for (uint32_t i = 0; i < n_owned; i++) add(owned[i], c1);
	for (uint32_t c = c1; c-- > c_lo;)
	{
		std::vector<uint32_t> &bk = s.bucket[c - c_lo];
		std::sort(bk.begin(), bk.end());
		for (size_t k = 0; k < bk.size(); k++)
		{
			const uint32_t *b;
			const uint32_t nbod = g.bodies(bk[k], b);
			for (uint32_t j = 0; j < nbod; j++)
				if (s.stamp_p[b[j]] != serial)
				{
					s.halo.push_back(b[j]);
					add(b[j], c);
				}
		}
		after(c);
	}
}

// measured on MI355X with 1024-thread tiles (ns per slot of a tile; types not measured are scaled by their
// instruction count)
const double kSlotNs[PBDX_NUM_CONSTRAINT_TYPES] = { 0.8, 0.83, 1.6, 1.4, 1.46, 1.8, 2.2, 1.2, 1.25, 3.0, 3.2, 4.0, 5.0 };
const double kColourFixedNs = 900.0;
const double kFillNsPerParticle = 0.74;
const double kWriteBackNsPerParticle = 0.47;

uint32_t slot_bytes(const TypeView &v, int type)
{
	const TypeInfo *ti = type_info(type);
	return (ti->num_bodies == 2 ? 4u : 8u) + (uint32_t)num_planes(type, v.compact != 0) * 4u + (ti->xpbd ? 8u : 0u);
}

template <class F>
void parallel_for(uint32_t count, uint32_t threads, F &&fn)
{
	if (threads <= 1 || count <= 1)
	{
		for (uint32_t i = 0; i < count; i++) fn(i, 0u);
		return;
	}
	std::atomic<uint32_t> next(0);
	std::vector<std::thread> pool;
	for (uint32_t t = 0; t < threads; t++)
		pool.emplace_back([&, t]() {
			for (;;)
			{
				const uint32_t i = next.fetch_add(1);
				if (i >= count) break;
				fn(i, t);
			}
		});
	for (auto &th : pool) th.join();
}

// recursive coordinate bisection into k tiles of (nearly) equal particle count
void rcb(const float *x, uint32_t *perm, uint32_t count, uint32_t k)
{
	if (k <= 1 || count <= 1)
		return;
	float lo[3] = { 3.4e38f, 3.4e38f, 3.4e38f }, hi[3] = { -3.4e38f, -3.4e38f, -3.4e38f };
	for (uint32_t i = 0; i < count; i++)
		for (int d = 0; d < 3; d++)
		{
			const float v = x[3 * (size_t)perm[i] + d];
			if (v < lo[d]) lo[d] = v;
			if (v > hi[d]) hi[d] = v;
		}
	int axis = 0;
	for (int d = 1; d < 3; d++) if (hi[d] - lo[d] > hi[axis] - lo[axis]) axis = d;
	const uint32_t k_left = k / 2;
	const uint32_t n_left = (uint32_t)((uint64_t)count * k_left / k);
	std::nth_element(perm, perm + n_left, perm + count, [x, axis](uint32_t a, uint32_t b) {
		const float va = x[3 * (size_t)a + axis], vb = x[3 * (size_t)b + axis];
		return va < vb || (va == vb && a < b);
	});
	rcb(x, perm, n_left, k_left);
	rcb(x, perm + n_left, count - n_left, k - k_left);
}

// sizes of the tiles produced by rcb() in tile order (same split arithmetic)
void rcb_sizes(uint32_t count, uint32_t k, std::vector<uint32_t> &sizes)
{
	if (k <= 1 || count <= 1) { sizes.push_back(count); for (uint32_t i = 1; i < k; i++) sizes.push_back(0); return; }
	const uint32_t k_left = k / 2;
	const uint32_t n_left = (uint32_t)((uint64_t)count * k_left / k);
	rcb_sizes(n_left, k_left, sizes);
	rcb_sizes(count - n_left, k - k_left, sizes);
}

struct TileOut
{
	std::vector<uint32_t> gid;
	std::vector<FusedStep> steps;
	std::vector<uint16_t> idx;
	std::vector<float> params;
	std::vector<uint32_t> slot_cid;
	uint32_t lam_count = 0;
	uint32_t n_owned = 0;
	uint32_t slots = 0;
	uint64_t stream_bytes = 0;
	uint32_t tab_off = 0, tab_f4 = 0;       // dictionary form: the tile's table inside `params` (float offset) and its used size (16-byte units)
	// while the tile is being built: the distinct records of its dictionary candidates by type, and the candidate steps (slot -> record)
	std::unique_ptr<struct RecordTable> tables[PBDX_NUM_CONSTRAINT_TYPES];
	std::vector<struct DeferredStep> deferred;
};
struct DeferredStep { size_t step = 0; std::vector<uint32_t> entry; };

inline uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

// ---- dictionary form of wide, repetitive parameter records (FusedStep::dict) -------------------------
// distinct records of `words` dwords each, compared bit for bit (open addressing)
struct RecordTable
{
	uint32_t words;
	std::vector<uint32_t> keys;
	std::vector<uint32_t> cells;
	explicit RecordTable(uint32_t w) : words(w), cells(1024, 0xffffffffu) {}
	uint32_t size() const { return (uint32_t)(keys.size() / words); }
	static uint64_t hash(const uint32_t *r, uint32_t n)
	{
		uint64_t h = 1469598103934665603ull;
		for (uint32_t i = 0; i < n; i++) { h ^= r[i]; h *= 1099511628211ull; h ^= h >> 29; }
		return h;
	}
	void grow()
	{
		std::vector<uint32_t> bigger(cells.size() * 2, 0xffffffffu);
		for (uint32_t e = 0; e < size(); e++)
		{
			size_t c = hash(&keys[(size_t)e * words], words) & (bigger.size() - 1);
			while (bigger[c] != 0xffffffffu) c = (c + 1) & (bigger.size() - 1);
			bigger[c] = e;
		}
		cells.swap(bigger);
	}
	uint32_t find_or_add(const uint32_t *rec)
	{
		size_t c = hash(rec, words) & (cells.size() - 1);
		while (cells[c] != 0xffffffffu)
		{
			if (!memcmp(&keys[(size_t)cells[c] * words], rec, words * 4)) return cells[c];
			c = (c + 1) & (cells.size() - 1);
		}
		const uint32_t e = size();
		cells[c] = e;
		keys.insert(keys.end(), rec, rec + words);
		if ((size_t)size() * 2 > cells.size()) grow();
		return e;
	}
};
// room a tile's table gets in the parameter stream (floats): the whole LDS budget of a table -- the congruent copies of an instanced plan share the
// prototype's layout, and their records (rest geometry of a translated mesh, rounded elsewhere) differ from the prototype's in their last bits
inline uint32_t dict_area_floats() { return kDictTableF4 * 4u; }

// The dictionaries of one tile.  `steps`: the tile's steps; `candidate(si)`: may step si take the form (planning: every step of a dict_type; replicating an
// instanced plan: the steps the prototype converted); `record(si, q, out)`: the streamed planes of slot q of step si, plane order.  On success:
// entry_of[si][q] = the slot's offset in the table (16-byte units), `table` = the table (floats, used_f4 * 4), converted[si] = 1.  A type is converted
// if its records fit what is left of `capacity_f4` and repeat at least four times on average; `must`: fail (return false) instead of skipping a type.
template <class Candidate, class Record>
bool build_tile_dictionary(const FusedStep *steps, size_t nsteps, const TypeView *views, uint32_t capacity_f4, bool must, Candidate &&candidate, Record &&record,
	std::vector<std::vector<uint16_t>> &entry_of, std::vector<uint8_t> &converted, std::vector<float> &table, uint32_t &used_f4)
{
	entry_of.assign(nsteps, std::vector<uint16_t>());
	converted.assign(nsteps, 0);
	table.clear();
	used_f4 = 0;
	uint32_t rec[PBDX_MAX_PARAMS];
	for (int type = 0; type < PBDX_NUM_CONSTRAINT_TYPES; type++)
	{
		const uint32_t np = (uint32_t)num_planes(type, views[type].compact != 0), ef4 = dict_entry_f4(np);
		RecordTable rt(std::max(np, 1u));
		std::vector<std::vector<uint32_t>> ent(nsteps);
		uint64_t slots = 0;
		for (size_t si = 0; si < nsteps; si++)
		{
			if ((int)steps[si].type != type || !candidate(si)) continue;
			ent[si].resize(steps[si].count);
			for (uint32_t q = 0; q < steps[si].count; q++) { record(si, q, reinterpret_cast<float *>(rec)); ent[si][q] = rt.find_or_add(rec); }
			slots += steps[si].count;
		}
		if (!slots) continue;
		const uint64_t need = (uint64_t)rt.size() * ef4;
		if (used_f4 + need > capacity_f4 || (!must && slots < 4ull * rt.size()))
		{
			if (must) return false;
			continue;
		}
		table.resize((size_t)(used_f4 + need) * 4, 0.0f);
		for (uint32_t e = 0; e < rt.size(); e++) memcpy(&table[((size_t)used_f4 + (size_t)e * ef4) * 4], &rt.keys[(size_t)e * np], np * 4);
		for (size_t si = 0; si < nsteps; si++)
		{
			if (ent[si].empty()) continue;
			converted[si] = 1;
			entry_of[si].resize(ent[si].size());
			for (size_t q = 0; q < ent[si].size(); q++) entry_of[si][q] = (uint16_t)(used_f4 + ent[si][q] * ef4);
		}
		used_f4 += (uint32_t)need;
	}
	return true;
}
// floats a dictionary-form step occupies in the parameter stream: one uint16 per slot, whole 256-byte units
inline uint32_t dict_step_floats(uint32_t count) { return round_up((count + 1u) / 2u, 64u); }


// developer aid (PBDX_PLAN_VERBOSE): how much of the dictionary-eligible work took the form
void print_dictionary_coverage(const FusedPlan &plan)
{
	uint64_t cand = 0, conv = 0, tabs = 0, tiles = 0;
	uint32_t max_tab = 0;
	for (const FusedSegment &seg : plan.segs)
	{
		for (const FusedStep &st : seg.steps)
			if (dict_type((int)st.type)) { cand += st.count; if (st.dict) conv += st.count; }
		for (const FusedTile &t : seg.tiles) { tiles++; if (t.tab_f4) tabs++; max_tab = std::max(max_tab, t.tab_f4); }
	}
	fprintf(stderr, "[plan] dictionary form: %llu of %llu eligible slots, %llu of %llu (tile, segment) pairs carry a table, largest %u x 16 B\n",
		(unsigned long long)conv, (unsigned long long)cand, (unsigned long long)tabs, (unsigned long long)tiles, max_tab);
}

// planning: the tile is complete -- what its dictionary candidates become.  A type takes the form if its distinct records fit what is left of the
// table and repeat at least four times on average; its steps then hold one uint16 per slot (keep_streams -- the prototype of an instanced plan --: and
// the streamed block behind it, so that a copy whose records do not fit can fall back inside the same layout); otherwise they are streamed like any other.
void dictionary_finish(TileOut &o, const TypeView *views, bool vec, bool keep_streams)
{
	if (o.deferred.empty()) return;
	uint32_t used_f4 = 0, base_of[PBDX_NUM_CONSTRAINT_TYPES];
	bool converted[PBDX_NUM_CONSTRAINT_TYPES] = {};
	uint64_t slots_of[PBDX_NUM_CONSTRAINT_TYPES] = {};
	for (const DeferredStep &d : o.deferred) slots_of[o.steps[d.step].type] += o.steps[d.step].count;
	std::vector<float> table;
	for (int type = 0; type < PBDX_NUM_CONSTRAINT_TYPES; type++)
	{
		const RecordTable *rt = o.tables[type].get();
		if (!rt || !rt->size()) continue;
		const uint32_t np = rt->words, ef4 = dict_entry_f4(np);
		const uint64_t need = (uint64_t)rt->size() * ef4;
		if (used_f4 + need > kDictTableF4 || slots_of[type] < 4ull * rt->size()) continue;
		converted[type] = true;
		base_of[type] = used_f4;
		table.resize((size_t)(used_f4 + need) * 4, 0.0f);
		for (uint32_t e = 0; e < rt->size(); e++) memcpy(&table[((size_t)used_f4 + (size_t)e * ef4) * 4], &rt->keys[(size_t)e * np], np * 4);
		used_f4 += (uint32_t)need;
	}
	for (const DeferredStep &d : o.deferred)
	{
		FusedStep &st = o.steps[d.step];
		const RecordTable &rt = *o.tables[st.type];
		const uint32_t np = rt.words, groups64 = (st.count + 63) / 64;
		st.par_off = (uint32_t)o.params.size();
		if (converted[st.type])
		{
			st.dict = 1;
			o.params.resize(o.params.size() + dict_step_floats(st.count), 0.0f);
			uint16_t *ix = reinterpret_cast<uint16_t *>(&o.params[st.par_off]);
			for (uint32_t q = 0; q < st.count; q++) ix[q] = (uint16_t)(base_of[st.type] + d.entry[q] * dict_entry_f4(np));
			o.stream_bytes -= (uint64_t)st.count * (np * 4u - 2u);
			if (!keep_streams) continue;
		}
		// streamed block (a type that did not convert; or the fallback block of an instanced prototype), from the table's records
		const size_t off = o.params.size();
		o.params.resize(off + (size_t)groups64 * np * 64, 0.0f);
		for (uint32_t q = 0; q < st.count; q++)
		{
			const uint32_t *r = &rt.keys[(size_t)d.entry[q] * np];
			for (uint32_t pl = 0; pl < np; pl++) memcpy(&o.params[off + param_float_index(vec, np, pl, q)], &r[pl], 4);
		}
	}
	if (used_f4)
	{
		o.tab_off = (uint32_t)o.params.size();
		o.tab_f4 = used_f4;
		o.params.resize(o.params.size() + dict_area_floats(), 0.0f);
		memcpy(&o.params[o.tab_off], table.data(), table.size() * sizeof(float));
	}
	o.deferred.clear();
	for (auto &t : o.tables) t.reset();
}

} // namespace

void compute_type_view(int type, const std::vector<ParamSpan> &spans, TypeView &v)
{
	memset(&v, 0, sizeof(v));
	const TypeInfo *ti = type_info(type);
	if (!ti) return;
	const uint32_t np = ti->param_stride;
	bool seen = false, compact = true;
	uint32_t first[PBDX_MAX_PARAMS] = {};
	for (const ParamSpan &sp : spans)
	{
		if (!sp.count) continue;
		if (!seen) { memcpy(first, sp.params, np * 4); seen = true; }
		for (uint32_t i = 0; i < sp.count && compact; i++)
		{
			const uint32_t *row = reinterpret_cast<const uint32_t *>(sp.params) + (size_t)i * np;
			for (uint32_t k = 0; k < np; k++)
				if (((kCompactScalars[type] >> k) & 1u) && row[k] != first[k]) compact = false;
			if (is_bending_type(type))
				for (int c = 0; c < 4; c++)
					for (int r = c + 1; r < 4; r++)
						if (row[1 + c * 4 + r] != row[1 + r * 4 + c]) compact = false;
		}
	}
	v.compact = (seen && compact) ? 1u : 0u;
	if (v.compact)
		for (uint32_t k = 0; k < np; k++)
			if ((kCompactScalars[type] >> k) & 1u) memcpy(&v.u[k], &first[k], 4);
}

// Tile count when the caller does not fix the tile size.  One workgroup (= one tile) runs per CU at a time.
//  * large scenes: a whole number of "waves" of num_cus tiles, tiles as large as the LDS allows (fewer waves, less halo
//    redundancy): measured on the 64 x 200x200 ensemble, 512 tiles of 5 000 particles beat 768 x 3 333 by 10 % and 625 x 4 100 by
//    18 %.  About half of the LDS is needed for the halo of a 12-15 colour segment, hence at most ~5 200 owned particles.
//  * small scenes (fewer than 512 particles per CU): a colour step of a tile is one projection chain of a lone wave whatever its
//    slot count, so what a smaller tile saves is pass-boundary work (LDS fill, write-back, hand-off all scale with the particles
//    a tile stages) while the extra halo slots ride along for free -- as long as every tile still has a CU of its own.  Measured
//    (profiles/r03b_c3_tile_sweep.log, r03c_small_scene_tile_sweep.log): the 100 k-tet bar (23 331 particles) 46 tiles x 507:
//    0.712 ms, 183 x 128: 0.640, 234 x 100: 0.636, 257 x 91 (> 256 CUs): 1.20; 100x100 cloth 20 x 512: 0.358, 63 x 160: 0.324;
//    200x200 cloth 79 x 512: 0.376, 250 x 160: 0.343, 400 x 100: 0.63.  => ~128 particles per tile, never more tiles than CUs.
uint64_t default_tile_count(uint64_t n, uint32_t num_cus, uint32_t max_local)
{
	const uint64_t cus = std::max(1u, num_cus);
	const uint64_t t_max = std::max(512u, std::min(5200u, max_local / 2 + max_local / 50));
	if (n <= cus * 512u)
		return std::max<uint64_t>(1, std::min<uint64_t>(cus, (n + 127) / 128));
	return cus * ((n + cus * t_max - 1) / (cus * t_max));
}

bool build_fused_plan(uint32_t n, const float *x, const std::vector<PlanBatch> &batches,
	const PlanOptions &opt, FusedPlan &plan, std::string &why)
{
	const auto t_start = std::chrono::steady_clock::now();
	plan = FusedPlan();
	if (n == 0 || batches.empty()) { why = "empty schedule"; return false; }
	if (batches.size() >= 65535) { why = "too many batches"; return false; }
	const bool verbose_build = getenv("PBDX_PLAN_VERBOSE") != nullptr;
	auto lap_build = [&](const char *what) { if (verbose_build) fprintf(stderr, "[plan] %-24s %7.3f s since start\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count()); };

	Graph g;
	g.n = n;
	g.batches = &batches;
	g.batch_base.resize(batches.size() + 1);
	uint64_t total = 0;
	uint32_t ncol = 0;
	for (size_t b = 0; b < batches.size(); b++)
	{
		g.batch_base[b] = (uint32_t)total;
		total += batches[b].count;
		if (b && batches[b].colour < batches[b - 1].colour) { why = "batches not in colour order"; return false; }
		ncol = std::max(ncol, batches[b].colour + 1);
		if (!type_info(batches[b].type)) { why = "unknown constraint type"; return false; }
	}
	if (total >= 0xffffffffull) { why = "too many constraints"; return false; }
	g.batch_base[batches.size()] = (uint32_t)total;
	g.nc = (uint32_t)total;
	g.ncolours = ncol;
	g.cbatch.resize(g.nc);
	g.adj_off.assign((size_t)n + 2, 0);
	for (size_t b = 0; b < batches.size(); b++)
	{
		const uint32_t nbod = type_info(batches[b].type)->num_bodies;
		for (uint32_t i = 0; i < batches[b].count; i++)
		{
			g.cbatch[g.batch_base[b] + i] = (uint16_t)b;
			for (uint32_t j = 0; j < nbod; j++)
			{
				const uint32_t p = batches[b].idx[(size_t)i * nbod + j];
				if (p >= n) { why = "particle index out of range"; return false; }
				g.adj_off[p + 2]++;
			}
		}
	}
	for (size_t p = 2; p < g.adj_off.size(); p++) g.adj_off[p] += g.adj_off[p - 1];
	g.adj.resize(g.adj_off[n + 1]);
	for (size_t b = 0; b < batches.size(); b++)
	{
		const uint32_t nbod = type_info(batches[b].type)->num_bodies;
		for (uint32_t i = 0; i < batches[b].count; i++)
			for (uint32_t j = 0; j < nbod; j++)
				g.adj[g.adj_off[batches[b].idx[(size_t)i * nbod + j] + 1]++] = g.batch_base[b] + i;
	}
	g.adj_off.pop_back();     // adj_off[p] .. adj_off[p+1]

	lap_build("graph");
	// ---- parameter views: uniform over the whole schedule -> scalar ------------------------------
	for (int t = 0; t < PBDX_NUM_CONSTRAINT_TYPES; t++)
	{
		std::vector<ParamSpan> spans;
		for (const PlanBatch &pb : batches) if (pb.type == t && pb.count) spans.push_back({ pb.params, pb.count });
		compute_type_view(t, spans, plan.views[t]);
	}

	lap_build("parameter views");
	// ---- tiles ---------------------------------------------------------------------------------
	// (tile count: default_tile_count above)
	uint32_t T = opt.tile_particles, k;
	if (T == 0)
	{
		k = (uint32_t)default_tile_count(n, opt.num_cus, opt.sizing_local ? opt.sizing_local : opt.max_local);
	}
	else
	{
		T = std::min(T, opt.max_local);
		k = (n + T - 1) / T;
	}
	k = std::max(k, 1u);
	std::vector<uint32_t> perm(n);
	for (uint32_t i = 0; i < n; i++) perm[i] = i;
	rcb(x, perm.data(), n, k);
	std::vector<uint32_t> sizes;
	rcb_sizes(n, k, sizes);
	std::vector<uint32_t> tile_begin(k + 1, 0);
	for (uint32_t t = 0; t < k; t++) tile_begin[t + 1] = tile_begin[t] + sizes[t];
	plan.tile_of.resize(n);
	for (uint32_t t = 0; t < k; t++)
	{
		std::sort(perm.begin() + tile_begin[t], perm.begin() + tile_begin[t + 1]);
		for (uint32_t i = tile_begin[t]; i < tile_begin[t + 1]; i++) plan.tile_of[perm[i]] = t;
	}
	plan.num_tiles = k;
	plan.num_colours = ncol;
	plan.num_particles = n;
	plan.num_constraints = g.nc;
	plan.batch_base = g.batch_base;

	lap_build("tiles (bisection)");
	uint32_t threads = opt.threads ? opt.threads : std::min(32u, std::max(1u, std::thread::hardware_concurrency()));      // (32: the tile builds of the 1 M cloth scale to about there on the 256-CPU host)
	// (every thread's stamps are a pass over n + nc words: 32 MB per thread at 1 M particles / 6 M constraints -- initialised by the threads themselves;
	// done one after the other on the calling thread this was the largest serial piece of the whole build: 1.05 of 4.8 s on 8 cores)
	std::vector<Scratch> scratch(threads);
	parallel_for(threads, threads, [&](uint32_t i, uint32_t) { scratch[i].init(g); });
	lap_build("scratch");

	// ---- segment boundaries: dynamic programme over a sample of tiles ---------------------------
	// (developer aid: PBDX_PLAN_SLOT_SCALE / PBDX_PLAN_FIXED_NS / PBDX_PLAN_LAUNCH_NS rescale the time model to explore other
	// segmentations on the GPU; never set in production)
	// Small scenes (fewer than 512 particles per CU; ~128-particle tiles, at most one per CU): a colour step of a tile holds fewer slots than the workgroup has
	// lanes, so its duration is one projection chain of a lone wave whatever the slot count (step traces of the 100 k-tet bar:
	// 1.00 us for every step of 100-250 FEM slots; lone-wave VALU issue interval 4.5 cycles, scripts/microbench/valu_single.hip)
	// and redundant halo slots are nearly free: the per-slot cost is discounted, which makes segments longer (fewer tile-to-tile
	// hand-offs per sweep).  Measured on the bar: 5 -> 4 passes per sweep, -3 % (FEM), -3 % (XPBD distance + volume).
	const bool latency_bound = (uint64_t)n <= (uint64_t)std::max(1u, opt.num_cus) * 512u;
	const double slot_scale = getenv("PBDX_PLAN_SLOT_SCALE") ? atof(getenv("PBDX_PLAN_SLOT_SCALE")) : (latency_bound ? 0.3 : 1.0);
	const double fixed_ns = getenv("PBDX_PLAN_FIXED_NS") ? atof(getenv("PBDX_PLAN_FIXED_NS")) : kColourFixedNs;
	const double launch_ns = getenv("PBDX_PLAN_LAUNCH_NS") ? atof(getenv("PBDX_PLAN_LAUNCH_NS")) : opt.launch_cost_ns;
	const uint32_t maxlen = std::max(1u, std::min(opt.max_segment_colours, ncol));
	const uint32_t nsample = std::min(k, 8u);
	std::vector<uint32_t> sample(nsample);
	for (uint32_t i = 0; i < nsample; i++) sample[i] = (uint32_t)(((uint64_t)(2 * i + 1) * k) / (2 * nsample));
	// cost[c1][len-1] for segment [c1-len, c1)
	std::vector<double> seg_bytes((size_t)(ncol + 1) * maxlen, 0.0);
	std::vector<uint32_t> seg_maxlocal((size_t)(ncol + 1) * maxlen, 0u);
	{
		std::vector<std::vector<double>> tb(threads, std::vector<double>(seg_bytes.size(), 0.0));
		std::vector<std::vector<uint32_t>> tm(threads, std::vector<uint32_t>(seg_bytes.size(), 0u));
		parallel_for(ncol * nsample, threads, [&](uint32_t job, uint32_t th) {
			const uint32_t c1 = job / nsample + 1;
			const uint32_t t = sample[job % nsample];
			const uint32_t c_lo = c1 > maxlen ? c1 - maxlen : 0;
			Scratch &s = scratch[th];
			const uint32_t n_owned = tile_begin[t + 1] - tile_begin[t];
			double steps_ns = 0.0;
			closure(g, s, perm.data() + tile_begin[t], n_owned, c_lo, c1, [&](uint32_t c) {
				// time model of one tile (ns), from the per-step traces of the fused kernel: a fixed
				// cost per colour the tile takes part in (barrier + one projection chain), a per-slot cost by type
				// (the sweep is latency / issue bound, not byte bound), LDS fill and write-back per particle
				const std::vector<uint32_t> &bk = s.bucket[c - c_lo];
				if (!bk.empty()) steps_ns += fixed_ns;
				for (uint32_t cid : bk) steps_ns += slot_scale * kSlotNs[g.batch_of(cid).type];
				const uint32_t n_local = n_owned + (uint32_t)s.halo.size();
				const size_t e = (size_t)c1 * maxlen + (c1 - c - 1);
				tb[th][e] += steps_ns + kFillNsPerParticle * (opt.owned_stay_in_lds ? n_local - n_owned : n_local) + kWriteBackNsPerParticle * n_owned;
				tm[th][e] = std::max(tm[th][e], n_local);
			});
		});
		for (uint32_t th = 0; th < threads; th++)
			for (size_t e = 0; e < seg_bytes.size(); e++)
			{
				seg_bytes[e] += tb[th][e];
				seg_maxlocal[e] = std::max(seg_maxlocal[e], tm[th][e]);
			}
	}
	// tiles run concurrently, one per CU: a segment takes (mean tile time) x (waves of tiles) + one launch
	const double scale = (double)((k + opt.num_cus - 1) / std::max(1u, opt.num_cus)) / nsample;
	const uint32_t cap_dp = (uint32_t)(opt.max_local * (nsample < k ? 0.95 : 1.0));
	std::vector<double> best(ncol + 1, 1e300);
	std::vector<uint32_t> prev(ncol + 1, 0);
	best[0] = 0.0;
	for (uint32_t c1 = 1; c1 <= ncol; c1++)
		for (uint32_t len = 1; len <= maxlen && len <= c1; len++)
		{
			const size_t e = (size_t)c1 * maxlen + (len - 1);
			if (len > 1 && seg_maxlocal[e] > cap_dp) break;       // closures only grow with the length
			const double cst = best[c1 - len] + seg_bytes[e] * scale + launch_ns;
			if (cst < best[c1]) { best[c1] = cst; prev[c1] = c1 - len; }
		}
	lap_build("segment boundaries (DP)");
	std::vector<std::pair<uint32_t, uint32_t>> todo;
	for (uint32_t c = ncol; c > 0; c = prev[c]) todo.push_back({ prev[c], c });
	std::reverse(todo.begin(), todo.end());

	// ---- interior / boundary split of every tile's owned particles ------------------------------
	// A particle is BOUNDARY if some other tile stages it (its closure reaches it) in some segment.  The owned particles of a tile are
	// ordered interior first, boundary last (each part by particle id): in the persistent schedule, where a tile's owned particles stay in
	// LDS from pass to pass, only the boundary part has to reach memory between passes (FusedTile::wb_begin).  Costs one more closure per
	// (segment, tile); the order is the same for all segments.
	std::vector<uint32_t> wb_begin(k, 0);
	{
		std::vector<uint8_t> boundary(n, 0);
		for (size_t si = 0; si < todo.size(); si++)
		{
			const uint32_t c0 = todo[si].first, c1 = todo[si].second;
			parallel_for(k, threads, [&](uint32_t t, uint32_t th) {
				Scratch &s = scratch[th];
				closure(g, s, perm.data() + tile_begin[t], tile_begin[t + 1] - tile_begin[t], c0, c1, [](uint32_t) {});
				for (uint32_t h : s.halo) boundary[h] = 1;       // (benign race: every writer stores 1)
			});
		}
		for (uint32_t t = 0; t < k; t++)
		{
			uint32_t *first = perm.data() + tile_begin[t], *last = perm.data() + tile_begin[t + 1];
			uint32_t *mid = std::stable_partition(first, last, [&](uint32_t p) { return boundary[p] == 0; });      // both parts stay sorted by id
			const uint32_t n_owned = (uint32_t)(last - first);
			if (opt.bank_aware && getenv("PBDX_PLAN_CLASS_ORDER"))
			{
				// (developer experiment: numbered so that a slot's bank class follows the particle id, class_order)
				std::vector<uint32_t> tmp;
				class_order(first, (uint32_t)(mid - first), 0u, tmp);
				class_order(mid, (uint32_t)(last - mid), (uint32_t)(mid - first), tmp);
			}
			// the persistent fill re-reads the owned particles from index (n_owned & ~63) on (LDS-DMA granularity): they count as boundary
			wb_begin[t] = std::min((uint32_t)(mid - first), n_owned & ~63u);
		}
	}

	lap_build("boundary split");
	// ---- full build ----------------------------------------------------------------------------
	uint64_t slots_total = 0;
	for (size_t si = 0; si < todo.size(); si++)
	{
		const uint32_t c0 = todo[si].first, c1 = todo[si].second;
		std::vector<TileOut> outs(k);
		std::atomic<uint32_t> worst(0);
		parallel_for(k, threads, [&](uint32_t t, uint32_t th) {
			Scratch &s = scratch[th];
			TileOut &o = outs[t];
			const uint32_t *owned = perm.data() + tile_begin[t];
			o.n_owned = tile_begin[t + 1] - tile_begin[t];
			closure(g, s, owned, o.n_owned, c0, c1, [](uint32_t) {});
			std::sort(s.halo.begin(), s.halo.end());
			if (opt.bank_aware && getenv("PBDX_PLAN_CLASS_ORDER")) class_order(s.halo.data(), (uint32_t)s.halo.size(), o.n_owned, s.bank_order);
			const uint32_t n_local = o.n_owned + (uint32_t)s.halo.size();
			uint32_t w = worst.load();
			while (n_local > w && !worst.compare_exchange_weak(w, n_local)) {}
			if (n_local > opt.max_local) return;
			{
				uint32_t nsteps = 0;
				for (uint32_t c = c0; c < c1; c++)
				{
					const std::vector<uint32_t> &bk = s.bucket[c - c0];
					for (size_t q = 0; q < bk.size(); q++) if (q == 0 || g.cbatch[bk[q]] != g.cbatch[bk[q - 1]]) nsteps++;
				}
				if (nsteps > opt.max_tile_steps)
				{
					uint32_t big = opt.max_local + 1;     // reported through the same channel: forces a split
					uint32_t w2 = worst.load();
					while (big > w2 && !worst.compare_exchange_weak(w2, big)) {}
					return;
				}
			}
			o.gid.reserve(n_local);
			for (uint32_t i = 0; i < o.n_owned; i++) { s.local_of[owned[i]] = i; o.gid.push_back(owned[i]); }
			for (uint32_t i = 0; i < s.halo.size(); i++) { s.local_of[s.halo[i]] = o.n_owned + i; o.gid.push_back(s.halo[i]); }
			for (uint32_t c = c0; c < c1; c++)
			{
				const std::vector<uint32_t> &bk = s.bucket[c - c0];
				size_t a = 0;
				while (a < bk.size())
				{
					const uint32_t b = g.cbatch[bk[a]];
					size_t e = a;
					while (e < bk.size() && g.cbatch[bk[e]] == b) e++;
					const PlanBatch &pb = batches[b];
					const TypeInfo *ti = type_info(pb.type);
					const TypeView &v = plan.views[pb.type];
					FusedStep st;
					st.type = (uint32_t)pb.type;
					st.count = (uint32_t)(e - a);
					st.idx_off = (uint32_t)o.idx.size();
					st.par_off = (uint32_t)o.params.size();
					st.dict = 0;
					const uint32_t np_stream = (uint32_t)num_planes(pb.type, v.compact != 0);
					const uint32_t groups64 = (st.count + 63) / 64;
					st.lam_off = o.lam_count;
					st.barrier = (e == bk.size()) ? 1u : 0u;
					st.cid_off = (uint32_t)o.slot_cid.size();
					const uint32_t iw = ti->num_bodies == 2 ? 2 : 4;
					o.idx.resize(o.idx.size() + round_up(st.count * iw, 8), 0);
					// where parameter p of slot (64 g + l) goes: par_off + g * np * 64 + pa[p] + l * pb_[p] (param_float_index, resolved once per step)
					int plane_of[PBDX_MAX_PARAMS];
					uint32_t pa[PBDX_MAX_PARAMS], pb_[PBDX_MAX_PARAMS];
					for (uint32_t p = 0; p < ti->param_stride; p++)
					{
						plane_of[p] = param_streams(pb.type, v.compact != 0, (int)p) ? param_plane(pb.type, v.compact != 0, (int)p) : -1;
						if (plane_of[p] < 0) continue;
						pa[p] = (uint32_t)param_float_index(opt.vector_params, np_stream, (uint32_t)plane_of[p], 0);
						pb_[p] = (uint32_t)param_float_index(opt.vector_params, np_stream, (uint32_t)plane_of[p], 1) - pa[p];
					}
					// dictionary candidates (pbdx_plan.h dict_type) are not written yet: their records go into the tile's table of distinct records, and
					// what the step's part of the stream holds is decided when the tile is complete (dictionary_finish)
					const bool candidate = opt.dict_params && dict_type(pb.type) && np_stream >= 4;
					DeferredStep *def = nullptr;
					if (candidate)
					{
						if (!o.tables[pb.type]) o.tables[pb.type].reset(new RecordTable(np_stream));
						o.deferred.emplace_back();
						def = &o.deferred.back();
						def->step = o.steps.size();
						def->entry.resize(st.count);
					}
					else
						o.params.resize(o.params.size() + (size_t)groups64 * np_stream * 64, 0.0f);
					uint32_t rec[PBDX_MAX_PARAMS];
					// bank-aware order of the step's slots (pbdx_plan.h lds_bank_model): which lane projects which constraint
					const bool reorder = opt.bank_aware && st.count > 1u && !is_quad_type(pb.type) && pb.type != 11 /* strain tets: steps may run in quad form */;
					if (reorder)
					{
						const uint32_t nb = ti->num_bodies;
						s.bank_h.resize((size_t)st.count * nb);
						for (size_t q = a; q < e; q++)
						{
							const uint32_t i = bk[q] - g.batch_base[b];
							for (uint32_t j = 0; j < nb; j++) s.bank_h[(q - a) * nb + j] = s.local_of[pb.idx[(size_t)i * nb + j]];
						}
						const uint32_t *tkey = nullptr, *tcls = nullptr;
						if (candidate)
						{
							// the record of every slot first: lanes of a read group that fetch DIFFERENT records from the same bank class conflict
							s.bank_key.resize(st.count); s.bank_cls.resize(st.count);
							const uint32_t ef4 = dict_entry_f4(np_stream);
							for (size_t q = a; q < e; q++)
							{
								const float *src = pb.params + (size_t)(bk[q] - g.batch_base[b]) * ti->param_stride;
								for (uint32_t p = 0; p < ti->param_stride; p++) if (plane_of[p] >= 0) memcpy(&rec[plane_of[p]], &src[p], 4);
								const uint32_t en = o.tables[pb.type]->find_or_add(rec);
								s.bank_key[q - a] = en; s.bank_cls[q - a] = en * ef4;
							}
							tkey = s.bank_key.data(); tcls = s.bank_cls.data();
						}
						s.bank.run(st.count, nb, s.bank_h.data(), tkey, tcls, s.bank_order);
					}
					for (size_t q = a; q < e; q++)
					{
						const uint32_t cid = reorder ? bk[a + s.bank_order[q - a]] : bk[q];
						const uint32_t i = cid - g.batch_base[b];
						const uint32_t slot = (uint32_t)(q - a);
						for (uint32_t j = 0; j < ti->num_bodies; j++)
							o.idx[st.idx_off + slot * iw + j] = (uint16_t)s.local_of[pb.idx[(size_t)i * ti->num_bodies + j]];
						const float *src = pb.params + (size_t)i * ti->param_stride;
						if (candidate)
						{
							for (uint32_t p = 0; p < ti->param_stride; p++) if (plane_of[p] >= 0) memcpy(&rec[plane_of[p]], &src[p], 4);
							def->entry[slot] = o.tables[pb.type]->find_or_add(rec);
						}
						else
						{
							float *dst = &o.params[st.par_off + (size_t)(slot / 64) * np_stream * 64];
							const uint32_t l = slot % 64;
							for (uint32_t p = 0; p < ti->param_stride; p++) if (plane_of[p] >= 0) dst[pa[p] + l * pb_[p]] = src[p];
						}
						o.slot_cid.push_back(cid);
					}
					if (ti->xpbd) o.lam_count += round_up(st.count, 4);
					o.slots += st.count;
					o.stream_bytes += (uint64_t)st.count * slot_bytes(v, pb.type);
					o.steps.push_back(st);
					a = e;
				}
			}
			if (opt.dict_params) dictionary_finish(o, plan.views, opt.vector_params, opt.dict_keep_streams);
		});
		if (worst.load() > opt.max_local)
		{
			if (c1 - c0 <= 1) { why = "a single colour does not fit the LDS tile (particles or step descriptors): reduce tile_particles"; return false; }
			// the sampled estimate missed a larger tile: split this segment and retry both halves
			const uint32_t mid = (c0 + c1) / 2;
			todo[si] = { c0, mid };
			todo.insert(todo.begin() + si + 1, { mid, c1 });
			si--;
			continue;
		}
		lap_build("tiles of a segment built");
		plan.segs.emplace_back();
		FusedSegment &seg = plan.segs.back();
		seg.colour_begin = c0;
		seg.colour_end = c1;
		seg.vector_params = opt.vector_params;
		for (size_t b = 0; b < batches.size(); b++)
			if (batches[b].colour >= c0 && batches[b].colour < c1) { seg.constraints += batches[b].count; seg.type_mask |= 1u << batches[b].type; }
		seg.tiles.resize(k);
		// where every tile's part of the segment's streams goes (a prefix sum on the calling thread), then the copies by all threads: the streams of a
		// segment of the 1 M cloth are 140 MB, and appending them tile after tile to growing vectors was half a second per segment
		struct Base { uint64_t idx, par, gid, cid, step; uint32_t lam; };
		std::vector<Base> base(k + 1);
		base[0] = Base{ 0, 0, 0, 0, 0, 0 };
		for (uint32_t t = 0; t < k; t++)
		{
			const TileOut &o = outs[t];
			base[t + 1] = Base{ base[t].idx + o.idx.size(), base[t].par + o.params.size(), base[t].gid + round_up((uint32_t)o.gid.size(), 4), base[t].cid + o.slot_cid.size(),
				base[t].step + o.steps.size(), base[t].lam + o.lam_count };
			// streams are addressed with 32-bit BYTE offsets (buffer descriptors)
			if (base[t + 1].idx * 2 >= 0xfffffff0ull || base[t + 1].par * 4 >= 0xfffffff0ull || (uint64_t)base[t + 1].lam * 4 >= 0xfffffff0ull || base[t + 1].gid >= 0xfffffff0ull)
			{ why = "a segment stream exceeds 4 GiB"; return false; }
		}
		seg.idx.resize(base[k].idx); seg.params.resize(base[k].par); seg.gid.assign(base[k].gid, 0u); seg.slot_cid.resize(base[k].cid); seg.steps.resize(base[k].step);
		seg.lam_count = base[k].lam;
		parallel_for(k, threads, [&](uint32_t t, uint32_t) {
			const TileOut &o = outs[t];
			FusedTile &ft = seg.tiles[t];
			memset(&ft, 0, sizeof(ft));
			ft.step_begin = (uint32_t)base[t].step;
			ft.step_end = (uint32_t)base[t + 1].step;
			ft.n_local = (uint32_t)o.gid.size();
			ft.n_owned = o.n_owned;
			ft.wb_begin = wb_begin[t];
			ft.gid_off = (uint32_t)base[t].gid;
			ft.slots = o.slots;
			const uint32_t idx_base = (uint32_t)base[t].idx, par_base = (uint32_t)base[t].par, lam_base = base[t].lam, cid_base = (uint32_t)base[t].cid;
			ft.tab_off = o.tab_f4 ? (par_base + o.tab_off) / 4u : 0u;      // (every block of the stream is a whole number of 256-byte units)
			ft.tab_f4 = o.tab_f4;
			for (size_t q = 0; q < o.steps.size(); q++)
			{
				FusedStep st = o.steps[q];
				st.idx_off += idx_base; st.par_off += par_base; st.lam_off += lam_base; st.cid_off += cid_base;
				seg.steps[base[t].step + q] = st;
			}
			if (!o.idx.empty()) memcpy(&seg.idx[base[t].idx], o.idx.data(), o.idx.size() * sizeof(uint16_t));
			if (!o.params.empty()) memcpy(&seg.params[base[t].par], o.params.data(), o.params.size() * sizeof(float));
			if (!o.gid.empty()) memcpy(&seg.gid[base[t].gid], o.gid.data(), o.gid.size() * sizeof(uint32_t));
			if (!o.slot_cid.empty()) memcpy(&seg.slot_cid[base[t].cid], o.slot_cid.data(), o.slot_cid.size() * sizeof(uint32_t));
		});
		for (uint32_t t = 0; t < k; t++)
		{
			const TileOut &o = outs[t];
			seg.max_tab_f4 = std::max(seg.max_tab_f4, o.tab_f4);
			seg.max_local = std::max(seg.max_local, (uint32_t)o.gid.size());
			seg.slots += o.slots;
			seg.stream_bytes += o.stream_bytes;
		}
		slots_total += seg.slots;
		lap_build("segment assembled");
	}
	plan.redundancy = g.nc ? (double)slots_total / (double)g.nc : 1.0;
	plan.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
	if (getenv("PBDX_PLAN_VERBOSE")) print_dictionary_coverage(plan);
	return true;
}

void relayout_params(FusedSegment &seg, const TypeView *views, bool vector_params)
{
	if (seg.vector_params == vector_params) return;
	std::vector<float> tmp;
	for (const FusedStep &st : seg.steps)
	{
		const uint32_t np = (uint32_t)num_planes((int)st.type, views[st.type].compact != 0);
		if (!np || !st.count || st.dict) continue;      // (a dictionary-form step holds slot -> record offsets: the same in both forms)
		const size_t floats = (size_t)((st.count + 63) / 64) * np * 64;
		tmp.assign(seg.params.begin() + st.par_off, seg.params.begin() + st.par_off + floats);
		for (uint32_t g = 0; g < (st.count + 63) / 64; g++)
			for (uint32_t plane = 0; plane < np; plane++)
				for (uint32_t l = 0; l < 64; l++)
					seg.params[st.par_off + param_float_index(vector_params, np, plane, g * 64 + l)] = tmp[param_float_index(seg.vector_params, np, plane, g * 64 + l)];
	}
	seg.vector_params = vector_params;
}

} // namespace pbdx (the debug entries below are C symbols)

extern "C" int pbdx_debug_relayout_params(int type, int compact, uint32_t slots, int from_vector, float *block, uint32_t *planes_out)
{
	using namespace pbdx;
	if (!type_info(type) || !block) return PBDX_ERR_INVALID;
	TypeView views[16] = {};
	views[type].compact = compact ? 1 : 0;
	const uint32_t np = (uint32_t)num_planes(type, compact != 0);
	if (planes_out) *planes_out = np;
	FusedSegment seg;
	seg.vector_params = from_vector != 0;
	FusedStep st = {};
	st.type = (uint32_t)type; st.count = slots; st.par_off = 0;
	seg.steps.push_back(st);
	seg.params.assign(block, block + (size_t)((slots + 63) / 64) * np * 64);
	relayout_params(seg, views, from_vector == 0);
	std::copy(seg.params.begin(), seg.params.end(), block);
	return PBDX_OK;
}
extern "C" uint64_t pbdx_debug_param_float_index(int vector_params, uint32_t planes, uint32_t plane, uint32_t slot)
{
	return (uint64_t)pbdx::param_float_index(vector_params != 0, planes, plane, slot);
}

namespace pbdx {
namespace {
inline uint64_t mix(uint64_t a, uint64_t b)
{
	a ^= b + 0x9e3779b97f4a7c15ull + (a << 6) + (a >> 2);
	a *= 0xff51afd7ed558ccdull;
	a ^= a >> 33;
	return a;
}
}

bool check_fused_plan(uint32_t n, const std::vector<PlanBatch> &batches, const FusedPlan &plan, std::string &why)
{
	char msg[256];
	if (plan.num_particles != n) { why = "particle count mismatch"; return false; }
	// colour-sequential sweep
	std::vector<uint64_t> want(n);
	for (uint32_t p = 0; p < n; p++) want[p] = mix(0x1234, p);
	std::vector<uint32_t> base(batches.size() + 1, 0);
	for (size_t b = 0; b < batches.size(); b++) base[b + 1] = base[b] + batches[b].count;
	auto apply = [&](uint32_t cid, size_t b, std::vector<uint64_t> &val, const uint32_t *bod, uint32_t nbod) {
		(void)b;
		uint64_t h = mix(0xabcdef, cid);
		for (uint32_t j = 0; j < nbod; j++) h = mix(h, val[bod[j]]);
		for (uint32_t j = 0; j < nbod; j++) val[bod[j]] = mix(mix(h, j), val[bod[j]]);
	};
	for (size_t b = 0; b < batches.size(); b++)
	{
		const uint32_t nbod = type_info(batches[b].type)->num_bodies;
		for (uint32_t i = 0; i < batches[b].count; i++)
			apply(base[b] + i, b, want, batches[b].idx + (size_t)i * nbod, nbod);
	}
	// fused schedule
	std::vector<uint64_t> cur(n), nxt(n);
	for (uint32_t p = 0; p < n; p++) cur[p] = mix(0x1234, p);
	std::vector<uint64_t> local;
	std::vector<uint32_t> owner_count(n);
	uint32_t expect_colour = 0;
	for (const FusedSegment &seg : plan.segs)
	{
		if (seg.colour_begin != expect_colour) { why = "segments do not tile the colour range"; return false; }
		expect_colour = seg.colour_end;
		std::fill(owner_count.begin(), owner_count.end(), 0u);
		for (const FusedTile &t : seg.tiles)
		{
			if (t.n_local > 65535 || t.n_owned > t.n_local) { why = "bad tile sizes"; return false; }
			if (t.tab_f4 > kDictTableF4 || ((size_t)t.tab_off + t.tab_f4) * 4 > seg.params.size() || t.tab_f4 > seg.max_tab_f4) { why = "bad dictionary table"; return false; }
			local.resize(t.n_local);
			for (uint32_t i = 0; i < t.n_local; i++)
			{
				const uint32_t p = seg.gid[t.gid_off + i];
				if (p >= n) { why = "gid out of range"; return false; }
				local[i] = cur[p];
			}
			for (uint32_t s = t.step_begin; s < t.step_end; s++)
			{
				const FusedStep &st = seg.steps[s];
				const TypeInfo *ti = type_info((int)st.type);
				if (!ti) { why = "bad step type"; return false; }
				const uint32_t iw = ti->num_bodies == 2 ? 2 : 4;
				for (uint32_t q = 0; q < st.count; q++)
				{
					const uint32_t cid = seg.slot_cid[st.cid_off + q];
					const size_t b = std::upper_bound(base.begin(), base.end(), cid) - base.begin() - 1;
					if (batches[b].type != (int)st.type) { why = "slot type mismatch"; return false; }
					if (batches[b].colour < seg.colour_begin || batches[b].colour >= seg.colour_end) { why = "slot colour outside its segment"; return false; }
					// the slot's streamed parameters, whatever form the step's part of the stream has (planes / vector segments / dictionary): bit for bit the
					// constraint's own
					{
						const bool compact = plan.views[st.type].compact != 0;
						const uint32_t np = (uint32_t)num_planes((int)st.type, compact);
						const float *rec = batches[b].params + (size_t)(cid - base[b]) * ti->param_stride;
						const float *entry = nullptr;
						if (st.dict)
						{
							if (!dict_type((int)st.type) || !t.tab_f4) { why = "dictionary-form step without a table"; return false; }
							const uint16_t e = reinterpret_cast<const uint16_t *>(&seg.params[st.par_off])[q];
							if ((uint32_t)e + dict_entry_f4(np) > t.tab_f4) { why = "dictionary offset outside the tile's table"; return false; }
							entry = &seg.params[((size_t)t.tab_off + e) * 4];
						}
						for (uint32_t pk = 0; pk < ti->param_stride; pk++)
						{
							if (!param_streams((int)st.type, compact, (int)pk)) continue;
							const uint32_t plane = (uint32_t)param_plane((int)st.type, compact, (int)pk);
							const float got = entry ? entry[plane] : seg.params[st.par_off + param_float_index(seg.vector_params, np, plane, q)];
							if (memcmp(&got, &rec[pk], 4)) { snprintf(msg, sizeof(msg), "constraint %u: streamed parameter %u differs from the constraint's", cid, pk); why = msg; return false; }
						}
					}
					uint32_t lb[4];
					for (uint32_t j = 0; j < ti->num_bodies; j++)
					{
						lb[j] = seg.idx[st.idx_off + q * iw + j];
						if (lb[j] >= t.n_local) { why = "local index out of range"; return false; }
						if (seg.gid[t.gid_off + lb[j]] != batches[b].idx[(size_t)(cid - base[b]) * ti->num_bodies + j]) { why = "local index maps to the wrong particle"; return false; }
					}
					uint64_t h = mix(0xabcdef, cid);
					for (uint32_t j = 0; j < ti->num_bodies; j++) h = mix(h, local[lb[j]]);
					for (uint32_t j = 0; j < ti->num_bodies; j++) local[lb[j]] = mix(mix(h, j), local[lb[j]]);
				}
			}
			for (uint32_t i = 0; i < t.n_owned; i++)
			{
				const uint32_t p = seg.gid[t.gid_off + i];
				nxt[p] = local[i];
				owner_count[p]++;
			}
		}
		for (uint32_t p = 0; p < n; p++)
			if (owner_count[p] != 1) { snprintf(msg, sizeof(msg), "particle %u owned by %u tiles", p, owner_count[p]); why = msg; return false; }
		cur.swap(nxt);
	}
	if (expect_colour != plan.num_colours) { why = "segments do not cover all colours"; return false; }
	for (uint32_t p = 0; p < n; p++)
		if (cur[p] != want[p]) { snprintf(msg, sizeof(msg), "particle %u: fused schedule differs from the colour-sequential sweep", p); why = msg; return false; }
	return true;
}

// ------------------------------------------------------------------------------------------------
// instanced schedules
// ------------------------------------------------------------------------------------------------
bool check_instancing(uint32_t n_proto, uint32_t K, const std::vector<PlanBatch> &batches)
{
	if (K < 2 || !n_proto) return false;
	for (const PlanBatch &b : batches) if (b.count % K) return false;
	std::atomic<int> bad(0);
	const uint32_t threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
	parallel_for(K - 1, threads, [&](uint32_t kk, uint32_t) {
		const uint32_t k = kk + 1;
		if (bad.load()) return;
		for (const PlanBatch &b : batches)
		{
			const uint32_t nb = type_info(b.type)->num_bodies;
			const size_t len = (size_t)(b.count / K) * nb;
			const uint32_t *p0 = b.idx, *pk = b.idx + (size_t)k * len;
			const uint32_t shift = k * n_proto;
			for (size_t i = 0; i < len; i++)
				if (p0[i] >= n_proto || pk[i] != p0[i] + shift) { bad.store(1); return; }
		}
	});
	return !bad.load();
}

bool build_instanced_plan(uint32_t n_proto, uint32_t K, const float *x, const std::vector<PlanBatch> &batches,
	const PlanOptions &opt, FusedPlan &plan, std::string &why)
{
	const auto t_start = std::chrono::steady_clock::now();
	const bool verbose = getenv("PBDX_PLAN_VERBOSE") != nullptr;
	auto lap = [&](const char *what) { if (verbose) fprintf(stderr, "[plan] %-28s %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count()); };
	if (!check_instancing(n_proto, K, batches)) { why = "the schedule is not K congruent instances"; return false; }
	lap("congruence check");
	// views over ALL instances (a parameter that is uniform within the prototype need not be uniform across the copies)
	TypeView full_views[16];
	for (int t = 0; t < PBDX_NUM_CONSTRAINT_TYPES; t++)
	{
		std::vector<ParamSpan> spans;
		for (const PlanBatch &pb : batches) if (pb.type == t && pb.count) spans.push_back({ pb.params, pb.count });
		compute_type_view(t, spans, full_views[t]);
	}
	lap("views over all instances");
	// the prototype's share of the schedule and of the tiles
	std::vector<PlanBatch> proto(batches);
	for (PlanBatch &b : proto) b.count /= K;
	PlanOptions po = opt;
	po.dict_keep_streams = true;
	if (!po.tile_particles)
	{
		// tile count of the whole as the planner would choose it (whole waves of num_cus tiles at the largest tile the LDS
		// allows), divided among the instances
		const uint64_t k_all = default_tile_count((uint64_t)n_proto * K, opt.num_cus, opt.sizing_local ? opt.sizing_local : opt.max_local);
		const uint32_t k_proto = (uint32_t)std::max<uint64_t>(1, (k_all + K / 2) / K);
		po.tile_particles = std::min<uint32_t>(opt.max_local, (n_proto + k_proto - 1) / k_proto);
	}
	FusedPlan pp;
	if (!build_fused_plan(n_proto, x, proto, po, pp, why)) return false;
	for (int t = 0; t < PBDX_NUM_CONSTRAINT_TYPES; t++)
		if (pp.views[t].compact != full_views[t].compact) { why = "a shared parameter is uniform in the prototype but not across the instances"; return false; }

	lap("prototype planned");
	plan = FusedPlan();
	for (int t = 0; t < 16; t++) plan.views[t] = t < PBDX_NUM_CONSTRAINT_TYPES ? full_views[t] : pp.views[t];
	const uint32_t kt = pp.num_tiles;
	plan.num_tiles = kt * K;
	plan.num_colours = pp.num_colours;
	plan.num_particles = n_proto * K;
	plan.num_constraints = pp.num_constraints * K;
	plan.redundancy = pp.redundancy;
	plan.batch_base.resize(batches.size() + 1);
	{
		uint64_t total = 0;
		for (size_t b = 0; b < batches.size(); b++) { plan.batch_base[b] = (uint32_t)total; total += batches[b].count; }
		if (total >= 0xffffffffull) { why = "too many constraints"; return false; }
		plan.batch_base[batches.size()] = (uint32_t)total;
	}
	plan.tile_of.resize((size_t)n_proto * K);
	for (uint32_t k = 0; k < K; k++)
		for (uint32_t p = 0; p < n_proto; p++) plan.tile_of[(size_t)k * n_proto + p] = pp.tile_of[p] + k * kt;
	const uint32_t threads = opt.threads ? opt.threads : std::min(32u, std::max(1u, std::thread::hardware_concurrency()));      // (32: the tile builds of the 1 M cloth scale to about there on the 256-CPU host)
	plan.segs.resize(pp.segs.size());
	for (size_t si = 0; si < pp.segs.size(); si++)
	{
		std::atomic<uint64_t> streamed_instead(0);
		const FusedSegment &ps = pp.segs[si];
		FusedSegment &seg = plan.segs[si];
		const size_t n_idx = ps.idx.size(), n_par = ps.params.size(), n_gid = ps.gid.size(), n_cid = ps.slot_cid.size(), n_steps = ps.steps.size();
		if ((uint64_t)n_idx * K * 2 >= 0xfffffff0ull || (uint64_t)n_par * K * 4 >= 0xfffffff0ull || (uint64_t)ps.lam_count * K * 4 >= 0xfffffff0ull)
		{ why = "a segment stream exceeds 4 GiB"; return false; }
		seg.colour_begin = ps.colour_begin; seg.colour_end = ps.colour_end;
		seg.vector_params = ps.vector_params;
		seg.lam_count = ps.lam_count * K;
		seg.max_local = ps.max_local;
		seg.type_mask = ps.type_mask;
		seg.slots = ps.slots * K; seg.constraints = ps.constraints * K; seg.stream_bytes = ps.stream_bytes * K;
		seg.tiles.resize((size_t)kt * K);
		seg.steps.resize(n_steps * K);
		seg.idx.resize(n_idx * K);
		seg.params.resize(n_par * K);
		seg.gid.resize(n_gid * K);
		seg.slot_cid.resize(n_cid * K);
		parallel_for(K, threads, [&](uint32_t k, uint32_t) {
			for (uint32_t t = 0; t < kt; t++)
			{
				FusedTile ft = ps.tiles[t];
				ft.step_begin += (uint32_t)(k * n_steps); ft.step_end += (uint32_t)(k * n_steps);
				ft.gid_off += (uint32_t)(k * n_gid);
				if (ft.tab_f4) ft.tab_off += (uint32_t)(k * n_par / 4);       // (its used size: below, from this instance's records)
				seg.tiles[(size_t)k * kt + t] = ft;
			}
			memcpy(&seg.idx[k * n_idx], ps.idx.data(), n_idx * sizeof(uint16_t));
			for (size_t i = 0; i < n_gid; i++) seg.gid[k * n_gid + i] = ps.gid[i] + k * n_proto;
			// padding words of the parameter stream stay zero, like the prototype's
			if (n_par) memset(&seg.params[k * n_par], 0, n_par * sizeof(float));
			for (size_t s2 = 0; s2 < n_steps; s2++)
			{
				FusedStep st = ps.steps[s2];
				const int type = (int)st.type;
				const TypeInfo *ti = type_info(type);
				const bool compact = plan.views[type].compact != 0;
				const uint32_t np_stream = (uint32_t)num_planes(type, compact);
				const uint32_t par_off0 = st.par_off, cid_off0 = st.cid_off;
				st.idx_off += (uint32_t)(k * n_idx); st.par_off += (uint32_t)(k * n_par); st.lam_off += k * ps.lam_count; st.cid_off += (uint32_t)(k * n_cid);
				seg.steps[k * n_steps + s2] = st;
				if (!st.count) continue;
				// a step is a run of ONE batch (one colour, one type): resolve the batch and the plane table once
				const uint32_t cid_first = ps.slot_cid[cid_off0];
				const size_t b = (size_t)(std::upper_bound(pp.batch_base.begin(), pp.batch_base.end(), cid_first) - pp.batch_base.begin()) - 1;
				const uint32_t base_p = pp.batch_base[b], cnt_p = proto[b].count, base_full = plan.batch_base[b] + k * cnt_p;
				int plane_of[PBDX_MAX_PARAMS];
				for (uint32_t pk = 0; pk < ti->param_stride; pk++)
					plane_of[pk] = param_streams(type, compact, (int)pk) ? param_plane(type, compact, (int)pk) : -1;
				const float *recs = batches[b].params + (size_t)k * cnt_p * ti->param_stride;
				float *dst0 = np_stream ? &seg.params[k * n_par + par_off0] : nullptr;
				uint32_t *cid_dst = &seg.slot_cid[k * n_cid + cid_off0];
				for (uint32_t q = 0; q < st.count; q++)
				{
					const uint32_t i = ps.slot_cid[cid_off0 + q] - base_p;       // position in the prototype's batch
					cid_dst[q] = base_full + i;
					if (np_stream && !st.dict)
					{
						const float *rec = recs + (size_t)i * ti->param_stride;
						for (uint32_t pk = 0; pk < ti->param_stride; pk++)
							if (plane_of[pk] >= 0) dst0[param_float_index(seg.vector_params, np_stream, (uint32_t)plane_of[pk], q)] = rec[pk];
					}
				}
			}
			// dictionary-form steps: this instance's OWN tables (its records differ from the prototype's in their last bits), in the prototype's layout
			std::vector<std::vector<uint16_t>> entry_of;
			std::vector<uint8_t> converted;
			std::vector<float> table;
			for (uint32_t t = 0; t < kt; t++)
			{
				const FusedTile &pt = ps.tiles[t];
				if (!pt.tab_f4) continue;
				const FusedStep *psteps = &ps.steps[pt.step_begin];
				const size_t nst = pt.step_end - pt.step_begin;
				uint32_t used_f4 = 0;
				build_tile_dictionary(psteps, nst, plan.views, dict_area_floats() / 4u, false,
					[&](size_t si) { return psteps[si].dict != 0; },
					[&](size_t si, uint32_t q, float *out) {
						const FusedStep &st = psteps[si];
						const int type = (int)st.type;
						const TypeInfo *ti = type_info(type);
						const bool compact = plan.views[type].compact != 0;
						const uint32_t cid_p = ps.slot_cid[st.cid_off + q];
						const size_t b = (size_t)(std::upper_bound(pp.batch_base.begin(), pp.batch_base.end(), cid_p) - pp.batch_base.begin()) - 1;
						const float *rec = batches[b].params + ((size_t)k * proto[b].count + (cid_p - pp.batch_base[b])) * ti->param_stride;
						for (uint32_t pk = 0; pk < ti->param_stride; pk++)
							if (param_streams(type, compact, (int)pk)) out[param_plane(type, compact, (int)pk)] = rec[pk];
					}, entry_of, converted, table, used_f4);
				for (size_t si = 0; si < nst; si++)
				{
					const FusedStep &st0 = psteps[si];
					if (!st0.dict) continue;
					FusedStep &mine = seg.steps[k * n_steps + pt.step_begin + si];
					if (converted[si]) { memcpy(&seg.params[k * n_par + st0.par_off], entry_of[si].data(), (size_t)st0.count * 2); continue; }
					// this copy's records of the type do not fit a table (or do not repeat): streamed, in the block the prototype kept behind the index area
					const int type = (int)st0.type;
					const TypeInfo *ti = type_info(type);
					const bool compact = plan.views[type].compact != 0;
					const uint32_t np = (uint32_t)num_planes(type, compact);
					mine.dict = 0;
					mine.par_off += dict_step_floats(st0.count);
					float *dst0 = &seg.params[k * n_par + st0.par_off + dict_step_floats(st0.count)];
					for (uint32_t q = 0; q < st0.count; q++)
					{
						const uint32_t cid_p = ps.slot_cid[st0.cid_off + q];
						const size_t b = (size_t)(std::upper_bound(pp.batch_base.begin(), pp.batch_base.end(), cid_p) - pp.batch_base.begin()) - 1;
						const float *rec = batches[b].params + ((size_t)k * proto[b].count + (cid_p - pp.batch_base[b])) * ti->param_stride;
						for (uint32_t pk = 0; pk < ti->param_stride; pk++)
							if (param_streams(type, compact, (int)pk)) dst0[param_float_index(seg.vector_params, np, (uint32_t)param_plane(type, compact, (int)pk), q)] = rec[pk];
					}
					streamed_instead += (uint64_t)st0.count * (np * 4u - 2u);
				}
				if (!table.empty()) memcpy(&seg.params[k * n_par + (size_t)pt.tab_off * 4], table.data(), table.size() * sizeof(float));
				seg.tiles[(size_t)k * kt + t].tab_f4 = used_f4;
			}
		});
		for (const FusedTile &ft : seg.tiles) seg.max_tab_f4 = std::max(seg.max_tab_f4, ft.tab_f4);
		seg.stream_bytes += streamed_instead.load();
	}
	lap("replicated");
	if (verbose) print_dictionary_coverage(plan);
	plan.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
	return true;
}


// LDS bank model of a plan (pbdx_plan.h): cycles the endpoint gathers / scatters and the dictionary reads of ONE sweep take on the LDS array
void lds_bank_model(const FusedPlan &plan, uint32_t block, LdsBankModel &out)
{
	out = LdsBankModel();
	uint64_t lb_read[PBDX_NUM_CONSTRAINT_TYPES] = {}, lb_write[PBDX_NUM_CONSTRAINT_TYPES] = {}, cyc_read[PBDX_NUM_CONSTRAINT_TYPES] = {}, cyc_write[PBDX_NUM_CONSTRAINT_TYPES] = {},
		grp_read[PBDX_NUM_CONSTRAINT_TYPES] = {}, grp_write[PBDX_NUM_CONSTRAINT_TYPES] = {};
	const bool verbose = getenv("PBDX_PLAN_VERBOSE") != nullptr;
	(void)block;      // (a chunk is a multiple of 64 slots: the wave a slot runs in, slot / 64, and its lane, slot % 64, do not depend on the workgroup size)
	for (const FusedSegment &seg : plan.segs)
		for (const FusedTile &t : seg.tiles)
			for (uint32_t si = t.step_begin; si < t.step_end; si++)
			{
				const FusedStep &st = seg.steps[si];
				const TypeInfo *ti = type_info((int)st.type);
				if (is_quad_type((int)st.type) || st.type == 11u) continue;
				const uint32_t nb = ti->num_bodies, iw = nb == 2 ? 2u : 4u;
				const uint32_t np = (uint32_t)num_planes((int)st.type, plan.views[st.type].compact != 0), ef4 = dict_entry_f4(np);
				const uint16_t *ent = st.dict ? reinterpret_cast<const uint16_t *>(&seg.params[st.par_off]) : nullptr;
				if (verbose)
				{
					// what ANY order of this step's slots can reach with the tile's slot numbering as it is: a group cycle serves one slot per bank class
					for (uint32_t j = 0; j < nb; j++)
					{
						uint32_t h16[16] = {}, h8[8] = {};
						for (uint32_t q = 0; q < st.count; q++) { const uint32_t h = seg.idx[st.idx_off + (size_t)q * iw + j]; h16[h & 15u]++; h8[h & 7u]++; }
						lb_read[st.type] += std::max((st.count + 15u) / 16u, *std::max_element(h16, h16 + 16));
						lb_write[st.type] += std::max((st.count + 7u) / 8u, *std::max_element(h8, h8 + 8));
					}
				}
				const uint64_t r0 = out.read_cycles, w0c = out.write_cycles, rg0 = out.read_groups, wg0 = out.write_groups;
				for (uint32_t w0 = 0; w0 < st.count; w0 += 64u)
				{
					const uint32_t lanes = std::min(64u, st.count - w0);
					for (uint32_t j = 0; j < nb; j++)
					{
						uint8_t rc[4][16] = {}, wc[8][8] = {};
						bool rn[4] = {}, wn[8] = {};
						for (uint32_t l = 0; l < lanes; l++)
						{
							const uint32_t h = seg.idx[st.idx_off + (size_t)(w0 + l) * iw + j];
							rc[lds_read_group(l)][h & 15u]++; rn[lds_read_group(l)] = true;
							wc[lds_write_group(l)][h & 7u]++; wn[lds_write_group(l)] = true;
						}
						for (uint32_t gq = 0; gq < 4; gq++) if (rn[gq]) { out.read_groups++; out.read_cycles += *std::max_element(rc[gq], rc[gq] + 16); }
						for (uint32_t gq = 0; gq < 8; gq++) if (wn[gq]) { out.write_groups++; out.write_cycles += *std::max_element(wc[gq], wc[gq] + 8); }
					}
					if (ent)
					{
						// distinct records per bank class and read group (equal addresses are one broadcast access); the ef4 reads of a record are this pattern shifted
						for (uint32_t gq = 0; gq < 4; gq++)
						{
							uint32_t seen[16][8]; uint32_t ns[16] = {}; bool any = false;
							for (uint32_t l = 0; l < lanes; l++)
							{
								if (lds_read_group(l) != gq) continue;
								any = true;
								const uint32_t off = t.n_local + ent[w0 + l], cls = off & 15u;
								bool dup = false;
								for (uint32_t k = 0; k < ns[cls] && k < 8u; k++) if (seen[cls][k] == off) dup = true;
								if (!dup) { if (ns[cls] < 8u) seen[cls][ns[cls]] = off; ns[cls]++; }
							}
							if (any) { out.table_groups += ef4; out.table_cycles += (uint64_t)ef4 * *std::max_element(ns, ns + 16); }
						}
					}
				}
				cyc_read[st.type] += out.read_cycles - r0; cyc_write[st.type] += out.write_cycles - w0c; grp_read[st.type] += out.read_groups - rg0; grp_write[st.type] += out.write_groups - wg0;
			}
	if (verbose)
		for (int t = 0; t < PBDX_NUM_CONSTRAINT_TYPES; t++)
			if (grp_read[t])
				fprintf(stderr, "[lds model] type %2d: reads %llu groups, best order %llu, this order %llu; writes %llu groups, best order %llu, this order %llu\n", t,
					(unsigned long long)grp_read[t], (unsigned long long)lb_read[t], (unsigned long long)cyc_read[t],
					(unsigned long long)grp_write[t], (unsigned long long)lb_write[t], (unsigned long long)cyc_write[t]);
}

void build_persistent_deps(const FusedPlan &plan, PersistentDeps &out)
{
	const size_t nseg = plan.segs.size();
	const uint32_t k = plan.num_tiles;
	out.off.assign(nseg, std::vector<uint32_t>());
	out.tile.assign(nseg, std::vector<uint32_t>());
	// read-after-write: owners of a tile's halo in segment si
	std::vector<std::vector<std::vector<uint32_t>>> raw(nseg, std::vector<std::vector<uint32_t>>(k));
	for (size_t si = 0; si < nseg; si++)
	{
		const FusedSegment &seg = plan.segs[si];
		for (uint32_t t = 0; t < k; t++)
		{
			const FusedTile &ft = seg.tiles[t];
			std::vector<uint32_t> &r = raw[si][t];
			for (uint32_t i = ft.n_owned; i < ft.n_local; i++) r.push_back(plan.tile_of[seg.gid[ft.gid_off + i]]);
			std::sort(r.begin(), r.end());
			r.erase(std::unique(r.begin(), r.end()), r.end());
		}
	}
	for (size_t si = 0; si < nseg; si++)
	{
		// + write-after-read: the tiles that read this tile's particles one pass earlier (segment si - 1)
		const size_t sp = (si + nseg - 1) % nseg;
		std::vector<std::vector<uint32_t>> dep = raw[si];
		for (uint32_t u = 0; u < k; u++)
			for (uint32_t t : raw[sp][u]) dep[t].push_back(u);
		out.off[si].assign(k + 1, 0);
		for (uint32_t t = 0; t < k; t++)
		{
			std::vector<uint32_t> &d = dep[t];
			std::sort(d.begin(), d.end());
			d.erase(std::unique(d.begin(), d.end()), d.end());
			d.erase(std::remove(d.begin(), d.end(), t), d.end());
			out.tile[si].insert(out.tile[si].end(), d.begin(), d.end());
			out.off[si][t + 1] = (uint32_t)out.tile[si].size();
		}
	}
}

bool check_persistent_deps(const FusedPlan &plan, const PersistentDeps &deps, uint32_t passes, std::string &why, bool keep_owned, uint32_t workgroups)
{
	// keep_owned with `workgroups` < num_tiles (0 = one workgroup per tile): workgroup w walks its tiles w, w + workgroups, ... in ascending order
	// in even passes and in descending order in odd ones, so the LAST tile of a pass is the FIRST of the next: that tile keeps its owned particles
	// in LDS across the boundary (stages only its halo; wrote back only its boundary particles), every other tile of the workgroup is staged and
	// written back in full.  The workgroup's order is part of the model: a tile's fill is enabled only when its predecessor has been written back.
	// (bit 31 of `workgroups`: self-test of this check -- the WRONG rule, boundary-only write-back for the FIRST tile of a pass instead of the last,
	// which leaves the interior of an evicted tile in no buffer; must be rejected wherever an evicted tile has an interior)
	const bool wrong_turn = (workgroups & 0x80000000u) != 0;
	workgroups &= 0x7fffffffu;
	const uint32_t grid = (keep_owned && workgroups && workgroups < plan.num_tiles) ? workgroups : plan.num_tiles;
	const size_t nseg = plan.segs.size();
	const uint32_t k = plan.num_tiles, n = plan.num_particles;
	if (!nseg || !k || deps.off.size() != nseg || deps.tile.size() != nseg) { why = "persistent deps: shape"; return false; }
	for (size_t si = 0; si < nseg; si++)
		if (deps.off[si].size() != (size_t)k + 1 || deps.off[si][k] != deps.tile[si].size()) { why = "persistent deps: CSR"; return false; }
	// position of tile t in its workgroup's walk of pass p: tiles of workgroup w = t % grid are w + j grid, j = 0 .. m - 1
	auto tiles_of = [&](uint32_t w) { return (k - w + grid - 1u) / grid; };
	auto walk_pos = [&](uint32_t t, uint32_t p) { const uint32_t j = t / grid, m = tiles_of(t % grid); return (p & 1u) ? m - 1u - j : j; };
	auto first_of_pass = [&](uint32_t t, uint32_t p) { return walk_pos(t, p) == 0u; };
	auto last_of_pass = [&](uint32_t t, uint32_t p) { return walk_pos(t, p) + 1u == tiles_of(t % grid); };
	// scheduler strategies: 0 most advanced tile first, 1 least advanced first, 2 pseudo-random, 3.. hold tile (strategy - 3) back
	const uint32_t held_samples = std::min(k, 6u);
	for (uint32_t strategy = 0; strategy < 3 + held_samples; strategy++)
	{
		const uint32_t held = strategy >= 3 ? (uint32_t)(((uint64_t)(strategy - 3) * k) / held_samples) : 0xffffffffu;
		// version of every particle in the two position buffers: the pass that wrote it (-1 = initial state, -2 = never written)
		std::vector<int32_t> ver[2] = { std::vector<int32_t>(n, -1), std::vector<int32_t>(n, -2) };
		std::vector<uint32_t> done(k, 0);          // passes published
		std::vector<uint8_t> filled(k, 0);         // FILL of pass done[t] has happened, WRITE-BACK has not
		std::vector<uint32_t> in_lds(grid, 0xffffffffu);      // the tile whose owned particles a workgroup's LDS holds
		uint64_t rng = 0x9e3779b97f4a7c15ull + strategy;
		uint64_t remaining = (uint64_t)k * passes * 2;
		while (remaining)
		{
			// enabled events
			uint32_t pick = 0xffffffffu, pick_held = 0xffffffffu;
			uint32_t best_key = 0;
			uint32_t seen = 0;
			for (uint32_t t = 0; t < k; t++)
			{
				if (done[t] >= passes) continue;
				bool enabled = true;
				if (!filled[t] && done[t] > 0)
				{
					const uint32_t p = done[t], si = p % (uint32_t)nseg;
					for (uint32_t d = deps.off[si][t]; d < deps.off[si][t + 1] && enabled; d++)
						if (done[deps.tile[si][d]] < p) enabled = false;
				}
				if (enabled && !filled[t] && grid < k)
				{
					// the workgroup's walk: the tile before this one in the pass has been written back
					const uint32_t p = done[t], pos = walk_pos(t, p);
					if (pos > 0u)
					{
						const uint32_t m = tiles_of(t % grid), jprev = (p & 1u) ? m - pos : pos - 1u;     // the tile at walk position pos - 1
						if (done[(t % grid) + jprev * grid] < p + 1u) enabled = false;
					}
				}
				if (!enabled) continue;
				if (t == held) { pick_held = t; continue; }
				const uint32_t prog = done[t] * 2 + filled[t];
				seen++;
				bool take = false;
				if (pick == 0xffffffffu) take = true;
				else if (strategy == 1) take = prog < best_key;
				else if (strategy == 2) { rng = rng * 6364136223846793005ull + 1442695040888963407ull; take = (rng >> 33) % seen == 0; }
				else take = prog > best_key;       // most advanced first (also while a tile is held back)
				if (take) { pick = t; best_key = prog; }
			}
			if (pick == 0xffffffffu) pick = pick_held;     // the held tile only moves when nothing else can
			if (pick == 0xffffffffu) { why = "persistent deps: the schedule deadlocks"; return false; }
			const uint32_t t = pick, p = done[t], si = p % (uint32_t)nseg;
			const FusedSegment &seg = plan.segs[si];
			const FusedTile &ft = seg.tiles[t];
			if (!filled[t])
			{
				const std::vector<int32_t> &in = ver[p & 1u];
				// keep_owned (one workgroup per tile): after its first pass a tile stages only what it does not hold already -- its halo and,
				// for the LDS-DMA granularity, its last (n_owned mod 64) owned particles
				const bool stays = keep_owned && p > 0 && first_of_pass(t, p);
				if (stays && in_lds[t % grid] != t)
				{
					char buf[200];
					snprintf(buf, sizeof(buf), "persistent deps: tile %u pass %u expects its owned particles in LDS, the workgroup holds tile %u", t, p, in_lds[t % grid]);
					why = buf;
					return false;
				}
				in_lds[t % grid] = t;
				for (uint32_t i = stays ? (ft.n_owned & ~63u) : 0u; i < ft.n_local; i++)
				{
					const uint32_t g = seg.gid[ft.gid_off + i];
					if (in[g] != (int32_t)p - 1)
					{
						char buf[256];
						snprintf(buf, sizeof(buf), "persistent deps: tile %u pass %u (segment %u) reads particle %u of tile %u at version %d, expected %d (scheduler %u)",
							t, p, si, g, plan.tile_of[g], in[g], (int32_t)p - 1, strategy);
						why = buf;
						return false;
					}
				}
				filled[t] = 1;
			}
			else
			{
				std::vector<int32_t> &outv = ver[(p + 1) & 1u];
				// ... and a pass that is not the last one writes back only its boundary particles (FusedTile::wb_begin)
				for (uint32_t i = (keep_owned && p + 1 < passes && (wrong_turn ? first_of_pass(t, p) : last_of_pass(t, p))) ? ft.wb_begin : 0u; i < ft.n_owned; i++) outv[seg.gid[ft.gid_off + i]] = (int32_t)p;
				filled[t] = 0;
				done[t] = p + 1;
			}
			remaining--;
		}
	}
	return true;
}

} // namespace pbdx
