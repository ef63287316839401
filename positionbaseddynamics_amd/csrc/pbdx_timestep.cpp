// pbdx_timestep.cpp -- host mirror of PBD::TimeStepController for particle scenes.
//
// Same parameter names and defaults as the reference
// (Simulation/TimeStepController.cpp:23-32, 47-72); `step` keeps TimeStep::step's
// contract (host ParticleData in -> one step -> host ParticleData out) and packs
// the model's colour groups into the engine's per-(group,type) batches the first
// time and whenever the model's topology or parameters changed.
#include "pbdx_internal.h"
#include "pbdx_plan.h"
#include <string.h>
#include <atomic>
#include <memory>
#include <thread>
#include <algorithm>

using namespace pbdx;

struct pbdx_timestep
{
	pbdx_solver *solver = nullptr;   // created by the first call that needs the device (parameters can be set without a GPU)
	int device = 0;
	uint32_t sub_steps = 5, max_iterations = 1, max_iterations_v = 5;
	int velocity_update_method = 0;
	float gravity[3] = { 0.0f, -9.81f, 0.0f };
	float h = 0.005f;
	float time = 0.0f;
	// device image bookkeeping
	const pbdx_model *image_of = nullptr;
	uint64_t topo = ~0ull, params = ~0ull;
	bool schedule_valid = false;
	bool device_ahead = false;   // device state newer than the host model (resident stepping)
	uint64_t state_seen = ~0ull; // model->state_version of the host state the device image was built from / synced to
	uint32_t image_n = 0;        // particles in the device image
	// what the host acceleration array was last filled for (clear_accelerations): while model, size, masses (params_version), host writes
	// (state_version), topology and gravity are the same, the pass would store the values that are there -- 0.8 ms per call at 1 M particles
	struct AccelKey { uint64_t m = 0; uint32_t n = 0; uint64_t state = 0, params = 0, topo = 0; float g[3] = { 0, 0, 0 }; } accel;
};

namespace {

// the engine is created on first use; without a HIP device this fails with PBDX_ERR_NO_DEVICE (no CPU path)
int ensure_solver(pbdx_timestep *ts)
{
	if (ts->solver) return PBDX_OK;
	return pbdx_solver_create(&ts->solver, ts->device);
}

int upload_particles(pbdx_timestep *ts, pbdx_model *m)
{
	return pbdx_solver_set_particles(ts->solver, m->size(), m->x.data(), m->v.data(), m->old_x.data(), m->last_x.data(),
		m->mass.data(), m->inv_mass.data());
}

// Walk the model's colour groups ONCE, bucket each group by constraint type (creation order kept inside a bucket) and hand
// every non-empty (group, type) batch to `emit`.  An instanced model (pbdx_model_add_instances) emits the prototype's bucket
// for instance 0, 1, ... in one batch (that IS the creation order of the colour group); the rest data of the copies is
// evaluated here, straight into the batch arrays, by a few host threads (one slice of instances each).
template <class F> int for_each_batch(pbdx_model *m, F &&emit)
{
	int r = pbdx_model_init_constraint_groups(m);        // TimeStepController.cpp:256
	if (r) return r;
	const uint32_t K = m->inst_count;
	const uint32_t ncp = (uint32_t)m->constraints.size();
	std::vector<uint32_t> members[PBDX_NUM_CONSTRAINT_TYPES];
	std::vector<uint32_t> idx;
	std::vector<float> par;
	for (uint32_t g = 0; g < m->groups.size(); g++)
	{
		for (auto &v : members) v.clear();
		for (uint32_t ci : m->groups[g]) members[m->constraints[ci].type].push_back(ci);
		for (int type = 0; type < PBDX_NUM_CONSTRAINT_TYPES; type++)
		{
			const std::vector<uint32_t> &mem = members[type];
			if (mem.empty()) continue;
			const TypeInfo *ti = type_info(type);
			const uint32_t nb = ti->num_bodies, np = ti->param_stride;
			const size_t cnt = mem.size();
			idx.resize(cnt * K * nb);
			par.resize(cnt * K * np);
			std::atomic<int> bad(0);
			auto fill = [&](uint32_t k0, uint32_t k1)
			{
				HostConstraint c;
				for (uint32_t k = k0; k < k1; k++)
					for (size_t i = 0; i < cnt; i++)
					{
						if (!model_constraint(m, (uint64_t)k * ncp + mem[i], c)) { bad.store(1); continue; }
						memcpy(&idx[((size_t)k * cnt + i) * nb], c.bodies, nb * sizeof(uint32_t));
						memcpy(&par[((size_t)k * cnt + i) * np], c.params, np * sizeof(float));
					}
			};
			const uint32_t threads = (K > 1 && cnt * K > 65536) ? std::min<uint32_t>(std::min<uint32_t>(K, 16u), std::max(1u, std::thread::hardware_concurrency())) : 1u;
			if (threads <= 1) fill(0, K);
			else
			{
				std::vector<std::thread> pool;
				for (uint32_t t = 0; t < threads; t++) pool.emplace_back(fill, (uint32_t)((uint64_t)K * t / threads), (uint32_t)((uint64_t)K * (t + 1) / threads));
				for (std::thread &t : pool) t.join();
			}
			if (bad.load()) { set_error("an instance of the model is not congruent to its prototype (degenerate element)"); return PBDX_ERR_INVALID; }
			r = emit(g, type, (uint32_t)(cnt * K), idx, par);
			if (r) return r;
		}
	}
	return PBDX_OK;
}

int build_schedule(pbdx_timestep *ts, pbdx_model *m)
{
	int r = pbdx_solver_set_instancing(ts->solver, m->inst_particles, m->inst_count);
	if (r) return r;
	r = pbdx_solver_begin_schedule(ts->solver);
	if (r) return r;
	r = for_each_batch(m, [&](uint32_t g, int type, uint32_t count, const std::vector<uint32_t> &idx, const std::vector<float> &par) {
		return pbdx_solver_add_batch(ts->solver, g, type, count, idx.data(), par.data(), type_info(type)->param_stride);
	});
	if (r) return r;
	return pbdx_solver_end_schedule(ts->solver);
}

// Bring the device image up to date with the model.  Particle state: the host arrays are uploaded when they were
// written through the model API (state_version), when masses / parameters / topology changed, or on demand.  If the
// device is AHEAD of the host at that moment (resident stepping without sync_to_host), the arrays the host did NOT
// write are first pulled from the device, so that a parameter edit, a setMass or a partial host write (positions
// only, say) between resident steps never rolls the simulation back to a stale host mirror.
int refresh_image(pbdx_timestep *ts, pbdx_model *m, bool force_particles)
{
	{
		int r0 = ensure_solver(ts);
		if (r0) return r0;
	}
	const bool stale_topology = !ts->schedule_valid || ts->image_of != m || ts->topo != m->topology_version || !m->groups_initialized;
	const bool stale_params = ts->params != m->params_version;
	// dirty tracking: the host state was written through the model API since the last upload / sync
	const bool host_dirty = ts->state_seen != m->state_version;
	if (stale_topology || stale_params || force_particles || host_dirty)
	{
		if (ts->device_ahead && ts->image_of == m && ts->image_n && ts->image_n <= m->size())
		{
			// (particles appended on the host since then exist only there; the packed arrays take the device's first image_n entries)
			const uint32_t d = host_dirty ? m->dirty_arrays : 0u;
			int r = pbdx_solver_get_particles(ts->solver, ts->image_n, (d & 1u) ? nullptr : m->x.data(), (d & 4u) ? nullptr : m->v.data(),
				(d & 16u) ? nullptr : m->old_x.data(), (d & 32u) ? nullptr : m->last_x.data());
			if (r) return r;
		}
		int r = upload_particles(ts, m);
		if (r) return r;
		ts->image_n = m->size();
		ts->device_ahead = false;
		ts->state_seen = m->state_version;
		m->dirty_arrays = 0;
	}
	if (stale_topology || stale_params)
	{
		int r = build_schedule(ts, m);
		if (r) return r;
		ts->image_of = m; ts->topo = m->topology_version; ts->params = m->params_version; ts->schedule_valid = true;
	}
	return PBDX_OK;
}

void clear_accelerations(pbdx_timestep *ts, pbdx_model *m)
{
	// TimeStep::clearAccelerations  TimeStep.cpp:28-62: static particles keep their (stale) value
	pbdx_timestep::AccelKey &k = ts->accel;
	if (k.m == m->uid && k.n == m->size() && k.state == m->state_version && k.params == m->params_version && k.topo == m->topology_version &&
		memcmp(k.g, ts->gravity, sizeof(k.g)) == 0) return;
	k.m = m->uid; k.n = m->size(); k.state = m->state_version; k.params = m->params_version; k.topo = m->topology_version; memcpy(k.g, ts->gravity, sizeof(k.g));
	for (uint32_t i = 0; i < m->size(); i++)
		if (m->mass[i] != 0.0f) { m->a[3 * i] = ts->gravity[0]; m->a[3 * i + 1] = ts->gravity[1]; m->a[3 * i + 2] = ts->gravity[2]; }
}

} // namespace

extern "C" {

int pbdx_model_plan_check(pbdx_model *m, uint32_t tile_particles, uint32_t lds_particles, uint32_t max_segment_colours, pbdx_plan_info *out)
{
	if (!m) return PBDX_ERR_INVALID;
	struct Held { std::vector<uint32_t> idx; std::vector<float> par; };
	std::vector<std::unique_ptr<Held>> held;
	std::vector<PlanBatch> pbs;
	uint32_t colour = 0, last_group = 0xffffffffu;
	int r = for_each_batch(m, [&](uint32_t g, int type, uint32_t count, const std::vector<uint32_t> &idx, const std::vector<float> &par) {
		if (last_group != 0xffffffffu && g != last_group) colour++;
		last_group = g;
		held.emplace_back(new Held{ idx, par });
		pbs.push_back({ type, colour, count, held.back()->idx.data(), held.back()->par.data() });
		return (int)PBDX_OK;
	});
	if (r) return r;
	PlanOptions opt;
	opt.tile_particles = tile_particles;
	if (lds_particles) opt.max_local = lds_particles;
	if (max_segment_colours) opt.max_segment_colours = max_segment_colours;
	// (the dictionary form of the wide records is planned here too, so that check_fused_plan compares every slot's table entry with its constraint)
	opt.dict_params = !getenv("PBDX_NO_DICT") && opt.max_local > 2u * kDictTableF4;
	if (opt.dict_params) { opt.sizing_local = opt.max_local; opt.max_local -= kDictTableF4; }
	FusedPlan plan;
	std::string why;
	if (m->inst_count > 1)
	{
		// instanced model: the plan is one instance's, replicated; the checks below run on the replicated plan of the WHOLE
		if (!build_instanced_plan(m->inst_particles, m->inst_count, m->x.data(), pbs, opt, plan, why)) { set_error("instanced plan: %s", why.c_str()); return PBDX_ERR_INVALID; }
	}
	else if (!build_fused_plan(m->size(), m->x.data(), pbs, opt, plan, why)) { set_error("plan: %s", why.c_str()); return PBDX_ERR_INVALID; }
	if (!check_fused_plan(m->size(), pbs, plan, why)) { set_error("plan check: %s", why.c_str()); return PBDX_ERR_INVALID; }
	{
		// the persistent schedule's tile-to-tile dependency lists: asynchronous-execution check (three sweeps), and the
		// check of the check -- with one dependency removed the simulation must find the stale read
		PersistentDeps deps;
		build_persistent_deps(plan, deps);
		const uint32_t passes = 3 * (uint32_t)plan.segs.size();
		if (!check_persistent_deps(plan, deps, passes, why)) { set_error("plan check: %s", why.c_str()); return PBDX_ERR_INVALID; }
		PersistentDeps broken = deps;
		bool removed = false;
		for (uint32_t t = 1; t < plan.num_tiles && !removed; t++)
			for (size_t si = 0; si < broken.off.size(); si++)
			{
				std::vector<uint32_t> &lst = broken.tile[si];
				std::vector<uint32_t> &off = broken.off[si];
				for (uint32_t d = off[t]; d < off[t + 1]; d++)
					if (lst[d] == 0u)       // tile 0 is one of the tiles the checker holds back
					{
						lst.erase(lst.begin() + d);
						for (uint32_t q = t + 1; q <= plan.num_tiles; q++) off[q]--;
						removed = true;
						break;
					}
			}
		if (removed && check_persistent_deps(plan, broken, passes, why))
		{ set_error("plan check: the asynchronous-execution check accepted dependency lists with a missing entry"); return PBDX_ERR_INVALID; }
		// the same with one workgroup per tile: owned particles stay in LDS, a pass that is not the last writes back only its boundary particles
		// (FusedTile::wb_begin) -- and the check of THAT check: a tile that keeps a particle to itself which a neighbour stages must be caught
		if (!check_persistent_deps(plan, deps, passes, why, true)) { set_error("plan check (owned particles resident): %s", why.c_str()); return PBDX_ERR_INVALID; }
		// (the boundary marks are conservative: they are computed before a segment may still be split, and a split only shrinks the closures,
		// so a flagged particle need not be staged by anybody -- a tile whose withheld particles nobody reads is rightly accepted.  The check
		// of the check therefore tries tiles until one IS caught, and fails only if none of up to 32 candidates is.)
		uint32_t tried = 0; bool caught = false;
		for (uint32_t t = 0; t < plan.num_tiles && tried < 32 && !caught; t++)
		{
			const FusedTile &ft0 = plan.segs[0].tiles[t];
			const uint32_t all = ft0.n_owned & ~63u;
			if (ft0.wb_begin >= all) continue;                     // nothing more to withhold in this tile
			std::vector<uint32_t> keep(plan.segs.size());
			for (size_t si = 0; si < plan.segs.size(); si++) { keep[si] = plan.segs[si].tiles[t].wb_begin; plan.segs[si].tiles[t].wb_begin = all; }
			const bool accepted = check_persistent_deps(plan, deps, passes, why, true);
			for (size_t si = 0; si < plan.segs.size(); si++) plan.segs[si].tiles[t].wb_begin = keep[si];
			tried++;
			if (!accepted) caught = true;
		}
		if (tried && !caught && plan.num_tiles > 1) { set_error("plan check: tiles that write back none of their boundary particles went unnoticed (%u tried)", tried); return PBDX_ERR_INVALID; }
		// fewer workgroups than tiles (scenes that do not fit the LDS of the whole device at once): a workgroup walks its tiles forwards in even
		// passes and backwards in odd ones, the tile at the turn keeps its owned particles in LDS -- two and three tiles per workgroup.  (No
		// check of the check here: the walk itself orders tiles, so a removed list entry can be implied by it and rightly go unnoticed.)
		for (uint32_t per_wg = 2; per_wg <= 3 && plan.num_tiles >= per_wg; per_wg++)
		{
			const uint32_t wgs = (plan.num_tiles + per_wg - 1u) / per_wg;
			if (!check_persistent_deps(plan, deps, passes, why, true, wgs)) { set_error("plan check (%u tiles per workgroup, walked in alternating order): %s", per_wg, why.c_str()); return PBDX_ERR_INVALID; }
			// the check of this check: with the turn at the wrong end (boundary-only write-back for the FIRST tile of a pass) the interior of
			// an evicted tile reaches no buffer -- to be caught wherever a tile at either end of a workgroup's walk has an interior
			bool has_interior = false;
			for (uint32_t t = 0; t < plan.num_tiles && !has_interior; t++)
			{
				// (only the tiles at the two ends of a workgroup's walk are ever first in a pass)
				const uint32_t m = (plan.num_tiles - (t % wgs) + wgs - 1u) / wgs, j = t / wgs;
				has_interior = plan.segs[0].tiles[t].wb_begin > 0 && m >= 2u && (j == 0u || j + 1u == m);
			}
			if (has_interior && check_persistent_deps(plan, deps, passes, why, true, wgs | 0x80000000u))
			{ set_error("plan check (%u tiles per workgroup): a walk that keeps the wrong tile went unnoticed", per_wg); return PBDX_ERR_INVALID; }
		}
	}
	if (out)
	{
		memset(out, 0, sizeof(*out));
		out->built = 1;
		out->num_segments = (uint32_t)plan.segs.size();
		out->num_tiles = plan.num_tiles;
		out->num_colours = plan.num_colours;
		out->redundancy = plan.redundancy;
		out->build_seconds = plan.build_seconds;
		for (const FusedSegment &seg : plan.segs)
		{
			out->max_local = out->max_local > seg.max_local ? out->max_local : seg.max_local;
			out->stream_bytes_per_sweep += seg.stream_bytes;
			out->slots_per_sweep += seg.slots;
		}
	}
	return PBDX_OK;
}

// Developer aid (host only): the plan the engine would build for this model on a 256-CU device with the one-launch schedule, planned with and without
// the bank-aware slot order, both evaluated under the LDS bank model (pbdx_plan.h) and the bank-aware one executed symbolically (check_fused_plan).
// out[0..5] = read groups, read cycles, write groups, write cycles, table groups, table cycles of the plan AS BUILT (bank_aware as given);
// out[6] = slots per sweep, out[7] = plan build time in microseconds.
int pbdx_debug_plan_lds_model(pbdx_model *m, int bank_aware, int check, uint64_t out[8])
{
	if (!m || !out) return PBDX_ERR_INVALID;
	struct Held { std::vector<uint32_t> idx; std::vector<float> par; };
	std::vector<std::unique_ptr<Held>> held;
	std::vector<PlanBatch> pbs;
	uint32_t colour = 0, last_group = 0xffffffffu, mask = 0;
	int r = for_each_batch(m, [&](uint32_t g, int type, uint32_t count, const std::vector<uint32_t> &idx, const std::vector<float> &par) {
		if (last_group != 0xffffffffu && g != last_group) colour++;
		last_group = g;
		mask |= 1u << type;
		held.emplace_back(new Held{ idx, par });
		pbs.push_back({ type, colour, count, held.back()->idx.data(), held.back()->par.data() });
		return (int)PBDX_OK;
	});
	if (r) return r;
	// (the rules of ensure_plan, pbdx_solver.hip, for a 256-CU device with 160 KiB of LDS per CU and one workgroup per CU)
	PlanOptions opt;
	opt.num_cus = 256;
	opt.max_local = 10240u - 256u;
	const bool big = (uint64_t)m->size() > 256ull * 1024ull;
	if (big) { opt.launch_cost_ns = 1500.0; opt.owned_stay_in_lds = true; }
	const uint32_t light = (1u << PBDX_DISTANCE) | (1u << PBDX_DISTANCE_XPBD) | (1u << PBDX_ISOMETRIC_BENDING) | (1u << PBDX_ISOMETRIC_BENDING_XPBD) |
		(1u << PBDX_VOLUME) | (1u << PBDX_VOLUME_XPBD) | (1u << PBDX_DIHEDRAL);
	opt.vector_params = (mask & ~light) != 0 || (uint64_t)m->size() <= 256ull * 512ull;
	opt.dict_params = !getenv("PBDX_NO_DICT");
	if (opt.dict_params) { opt.sizing_local = opt.max_local; opt.max_local -= kDictTableF4; }
	opt.bank_aware = bank_aware != 0;
	FusedPlan plan;
	std::string why;
	if (m->inst_count > 1)
	{
		if (!build_instanced_plan(m->inst_particles, m->inst_count, m->x.data(), pbs, opt, plan, why)) { set_error("instanced plan: %s", why.c_str()); return PBDX_ERR_INVALID; }
	}
	else if (!build_fused_plan(m->size(), m->x.data(), pbs, opt, plan, why)) { set_error("plan: %s", why.c_str()); return PBDX_ERR_INVALID; }
	if (check && !check_fused_plan(m->size(), pbs, plan, why)) { set_error("plan check: %s", why.c_str()); return PBDX_ERR_INVALID; }
	LdsBankModel lm;
	lds_bank_model(plan, 1024, lm);
	out[0] = lm.read_groups; out[1] = lm.read_cycles; out[2] = lm.write_groups; out[3] = lm.write_cycles; out[4] = lm.table_groups; out[5] = lm.table_cycles;
	out[6] = 0;
	for (const FusedSegment &seg : plan.segs) out[6] += seg.slots;
	out[7] = (uint64_t)(plan.build_seconds * 1e6);
	return PBDX_OK;
}

int pbdx_timestep_create(pbdx_timestep **out, int device)
{
	if (!out) { set_error("pbdx_timestep_create: null out"); return PBDX_ERR_INVALID; }
	*out = nullptr;
	pbdx_timestep *ts = new (std::nothrow) pbdx_timestep();
	if (!ts) { set_error("out of memory"); return PBDX_ERR_ALLOC; }
	ts->device = device;
	*out = ts;
	return PBDX_OK;
}

void pbdx_timestep_destroy(pbdx_timestep *ts)
{
	if (!ts) return;
	pbdx_solver_destroy(ts->solver);
	delete ts;
}

int pbdx_timestep_set_param(pbdx_timestep *ts, int id, int64_t value)
{
	if (!ts) return PBDX_ERR_INVALID;
	switch (id)
	{
	case PBDX_TS_NUM_SUB_STEPS: ts->sub_steps = value < 1 ? 1u : (uint32_t)value; break;          // setMinValue(1)
	case PBDX_TS_MAX_ITERATIONS: ts->max_iterations = value < 1 ? 1u : (uint32_t)value; break;    // setMinValue(1)
	case PBDX_TS_MAX_ITERATIONS_V: ts->max_iterations_v = value < 0 ? 0u : (uint32_t)value; break;
	case PBDX_TS_VELOCITY_UPDATE_METHOD:
		if (value != 0 && value != 1) { set_error("velocityUpdateMethod must be 0 or 1"); return PBDX_ERR_INVALID; }
		ts->velocity_update_method = (int)value; break;
	default: set_error("unknown time step parameter %d", id); return PBDX_ERR_INVALID;
	}
	return PBDX_OK;
}

int64_t pbdx_timestep_get_param(const pbdx_timestep *ts, int id)
{
	if (!ts) return -1;
	switch (id)
	{
	case PBDX_TS_NUM_SUB_STEPS: return ts->sub_steps;
	case PBDX_TS_MAX_ITERATIONS: return ts->max_iterations;
	case PBDX_TS_MAX_ITERATIONS_V: return ts->max_iterations_v;
	case PBDX_TS_VELOCITY_UPDATE_METHOD: return ts->velocity_update_method;
	default: return -1;
	}
}

const char *pbdx_timestep_param_name(int id)
{
	static const char *names[] = { "subSteps", "maxIterations", "maxIterationsV", "velocityUpdateMethod" };
	return (id >= 0 && id < 4) ? names[id] : "";
}

int pbdx_timestep_set_gravity(pbdx_timestep *ts, const float g[3])
{
	if (!ts || !g) return PBDX_ERR_INVALID;
	memcpy(ts->gravity, g, sizeof(ts->gravity));
	return PBDX_OK;
}
int pbdx_timestep_set_time_step_size(pbdx_timestep *ts, float h) { if (!ts) return PBDX_ERR_INVALID; ts->h = h; return PBDX_OK; }
float pbdx_timestep_get_time_step_size(const pbdx_timestep *ts) { return ts ? ts->h : 0.0f; }
float pbdx_timestep_get_time(const pbdx_timestep *ts) { return ts ? ts->time : 0.0f; }
int pbdx_timestep_reset(pbdx_timestep *ts) { if (!ts) return PBDX_ERR_INVALID; ts->time = 0.0f; ts->schedule_valid = false; ts->device_ahead = false; ts->state_seen = ~0ull; return PBDX_OK; }
int pbdx_timestep_invalidate(pbdx_timestep *ts) { if (!ts) return PBDX_ERR_INVALID; ts->schedule_valid = false; return PBDX_OK; }
pbdx_solver *pbdx_timestep_solver(pbdx_timestep *ts) { if (!ts) return nullptr; (void)ensure_solver(ts); return ts->solver; }

int pbdx_timestep_sync_to_host(pbdx_timestep *ts, pbdx_model *m)
{
	if (!ts || !m) return PBDX_ERR_INVALID;
	if (!ts->solver) { set_error("sync_to_host: nothing has been stepped on the device yet"); return PBDX_ERR_INVALID; }
	int r = pbdx_solver_get_particles(ts->solver, m->size(), m->x.data(), m->v.data(), m->old_x.data(), m->last_x.data());
	if (r) return r;
	ts->device_ahead = false;
	ts->state_seen = m->state_version;     // host mirror == device state
	m->dirty_arrays = 0;
	return PBDX_OK;
}

int pbdx_timestep_sync_from_host(pbdx_timestep *ts, pbdx_model *m)
{
	if (!ts || !m) return PBDX_ERR_INVALID;
	return refresh_image(ts, m, /*force_particles=*/true);
}

int pbdx_timestep_step(pbdx_timestep *ts, pbdx_model *m)
{
	if (!ts || !m) return PBDX_ERR_INVALID;
	// host ParticleData is authoritative (its arrays are freely writable through the model API)
	int r = refresh_image(ts, m, /*force_particles=*/true);
	if (r) return r;
	clear_accelerations(ts, m);
	r = pbdx_solver_step(ts->solver, ts->h, ts->sub_steps, ts->max_iterations, ts->velocity_update_method, ts->gravity, 1);
	if (r) return r;
	ts->time = ts->time + ts->h;                            // TimeStepController.cpp:239
	return pbdx_timestep_sync_to_host(ts, m);
}

int pbdx_timestep_project(pbdx_timestep *ts, pbdx_model *m, uint32_t iterations)
{
	if (!ts || !m) return PBDX_ERR_INVALID;
	int r = refresh_image(ts, m, /*force_particles=*/true);
	if (r) return r;
	r = pbdx_solver_project(ts->solver, ts->h / (float)ts->sub_steps, iterations);
	if (r) return r;
	return pbdx_solver_get_particles(ts->solver, m->size(), m->x.data(), nullptr, nullptr, nullptr);
}

int pbdx_timestep_step_resident(pbdx_timestep *ts, pbdx_model *m, uint32_t num_steps)
{
	if (!ts || !m) return PBDX_ERR_INVALID;
	int r = refresh_image(ts, m, /*force_particles=*/!ts->device_ahead && (ts->image_of != m));
	if (r) return r;
	clear_accelerations(ts, m);
	r = pbdx_solver_step(ts->solver, ts->h, ts->sub_steps, ts->max_iterations, ts->velocity_update_method, ts->gravity, num_steps);
	if (r) return r;
	for (uint32_t i = 0; i < num_steps; i++) ts->time = ts->time + ts->h;
	ts->device_ahead = true;
	return PBDX_OK;
}

} // extern "C"
