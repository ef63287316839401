// pbdx_ensemble.cpp -- the ensemble of independent scene instances over several HIP devices of ONE process (missing 1).
//
// A single sheet / bar is one connected colour-sequential Gauss-Seidel problem and does not shard (a split would need a halo exchange per colour and
// iteration; the reference has no counterpart: Simulation/TimeStepController.cpp:75-241 is one process on one model).  What shards is a model that holds
// K congruent independent instances (pbdx_model_add_instances): contiguous blocks of instances (pbdx_ensemble_shard), one block per device, NO exchange
// on the data path.  bench.py / positionbaseddynamics_amd/ensemble.py run that as one process per GPU under torch.distributed (RCCL carries barrier,
// times and checksums).  This file is the same thing for a C or C++ host in a single process: one engine (pbdx_timestep + its pbdx_solver, stream and
// device image) per listed device, every device stepping its block concurrently -- the engines' entry points select their device and restore the
// caller's (pbdx_device.h).  Every shard but the first has a RESIDENT host thread (created with the ensemble, parked on a condition variable): a step
// wakes them, each enqueues its device's work at once and waits for its own stream only (round 5 started one thread per device per call: 0.1-0.3 ms
// of thread creation in front of every step).  A single process needs no collective at all: what RCCL reduces across processes (projection counts,
// times, checksums) is summed on the host here; a host of SEVERAL processes has pbdx_comm_* (pbdx_comm.cpp: RCCL loaded at run time).
#include "pbdx_internal.h"
#include <string.h>
#include <string>
#include <thread>
#include <vector>
#include <chrono>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <memory>
#include <unistd.h>

using namespace pbdx;

struct pbdx_ensemble
{
	struct Shard
	{
		int device = 0;
		uint64_t begin = 0, end = 0;         // instances [begin, end) of the whole model
		pbdx_timestep *ts = nullptr;
		pbdx_model *model = nullptr;         // the block as a model of its own (instance `begin` is its prototype)
		double last_step_ms = 0.0;           // host wall clock of the shard's last step call
	};
	// one resident host thread per shard after the first (the calling thread takes shard 0)
	struct Worker
	{
		std::thread th;
		std::mutex mu;
		std::condition_variable cv;
		std::function<void()> job;
		bool has_job = false, quit = false;
	};
	std::vector<Shard> shards;
	std::vector<std::unique_ptr<Worker>> workers;      // workers[i] serves shard i + 1
	pid_t owner_pid = 0;                     // the threads exist in the process that created them only (after a fork the child runs its shards in turn)
	// the model of the last successful pbdx_ensemble_set_model: identified by its never-reused uid and the versions its blocks were cut from -- NOT by
	// address (the blocks are copies, the caller may destroy or move the model; an address can be reused)
	uint64_t whole_uid = 0;
	bool have_model = false;
	uint64_t whole_topo = ~0ull, whole_params = ~0ull, whole_state = ~0ull;
	double last_step_ms = 0.0;               // host wall clock of the last pbdx_ensemble_step (all devices)
};

namespace {

// the block [b, e) of an instanced model as a model of its own: particle state sliced, instance b as the prototype -- its constraints with the rest data
// the WHOLE model evaluates for instance b (model_constraint), so that every record of the block is bit for bit the whole model's
pbdx_model *slice_model(const pbdx_model *m, uint64_t b, uint64_t e, std::string &why)
{
	pbdx_model *s = nullptr;
	if (pbdx_model_create(&s) != PBDX_OK) { why = pbdx_last_error(); return nullptr; }
	const uint32_t np = m->inst_count > 1 ? m->inst_particles : m->size();
	const size_t p0 = (size_t)b * np, p1 = (size_t)e * np;
	auto cut1 = [&](const ParticleArray &a, ParticleArray &d) { d.assign(a.begin() + p0, a.begin() + p1); };
	auto cut3 = [&](const ParticleArray &a, ParticleArray &d) { d.assign(a.begin() + 3 * p0, a.begin() + 3 * p1); };
	cut1(m->mass, s->mass); cut1(m->inv_mass, s->inv_mass);
	cut3(m->x0, s->x0); cut3(m->x, s->x); cut3(m->v, s->v); cut3(m->a, s->a); cut3(m->old_x, s->old_x); cut3(m->last_x, s->last_x);
	s->tri_models = m->tri_models;
	s->tet_models = m->tet_models;
	s->groups = m->groups;
	s->groups_initialized = m->groups_initialized;
	const size_t nc = m->constraints.size();
	s->constraints.resize(nc);
	for (size_t i = 0; i < nc; i++)
	{
		HostConstraint c;
		if (!model_constraint(m, (uint64_t)b * nc + i, c)) { why = "an element of the block's first instance is degenerate"; pbdx_model_destroy(s); return nullptr; }
		for (uint32_t j = 0; j < type_info(c.type)->num_bodies; j++) c.bodies[j] -= (uint32_t)p0;
		s->constraints[i] = c;
	}
	s->inst_count = (uint32_t)(e - b);
	s->inst_particles = s->inst_count > 1 ? np : 0;
	if (s->inst_count > 1)
	{
		s->inst_offset.resize((size_t)3 * s->inst_count);
		for (uint64_t k = b; k < e; k++)
			for (int d = 0; d < 3; d++) s->inst_offset[3 * (k - b) + d] = m->inst_offset[3 * k + d] - m->inst_offset[3 * b + d];
	}
	s->topology_version = 1; s->params_version = 1; s->state_version = 1;
	return s;
}

// run fn(shard index) for every shard at once: the resident workers take shards 1 .., the calling thread shard 0; collects the first failure with its
// message.  In a forked child (the workers were not inherited) the shards run in turn on the calling thread.
template <class F> int for_all_shards(pbdx_ensemble *e, F &&fn)
{
	const size_t n = e->shards.size();
	std::vector<int> rc(n, PBDX_OK);
	std::vector<std::string> msg(n);
	auto body = [&](size_t i)
	{
		clear_error();
		rc[i] = fn(i);
		if (rc[i] != PBDX_OK) msg[i] = pbdx_last_error();      // (the message is thread-local: carried back by hand)
	};
	const bool team = e->owner_pid == getpid() && e->workers.size() + 1 == n;
	if (team)
	{
		std::mutex done_mu;
		std::condition_variable done_cv;
		size_t pending = n - 1;
		for (size_t i = 1; i < n; i++)
		{
			pbdx_ensemble::Worker &w = *e->workers[i - 1];
			std::lock_guard<std::mutex> lk(w.mu);
			w.job = [&, i] { body(i); std::lock_guard<std::mutex> dl(done_mu); if (--pending == 0) done_cv.notify_one(); };
			w.has_job = true;
			w.cv.notify_one();
		}
		if (n) body(0);
		std::unique_lock<std::mutex> dl(done_mu);
		done_cv.wait(dl, [&] { return pending == 0; });
	}
	else
		for (size_t i = 0; i < n; i++) body(i);
	for (size_t i = 0; i < n; i++)
		if (rc[i] != PBDX_OK) { set_error("ensemble shard %zu (device %d): %s", i, e->shards[i].device, msg[i].c_str()); return rc[i]; }
	return PBDX_OK;
}

void worker_main(pbdx_ensemble::Worker *w)
{
	for (;;)
	{
		std::function<void()> job;
		{
			std::unique_lock<std::mutex> lk(w->mu);
			w->cv.wait(lk, [&] { return w->has_job || w->quit; });
			if (w->quit) return;
			job.swap(w->job);
			w->has_job = false;
		}
		job();
	}
}

void drop_shards(pbdx_ensemble *e, bool keep_timesteps)
{
	for (pbdx_ensemble::Shard &s : e->shards)
	{
		if (s.model) { pbdx_model_destroy(s.model); s.model = nullptr; }
		if (!keep_timesteps && s.ts) { pbdx_timestep_destroy(s.ts); s.ts = nullptr; }
		s.begin = s.end = 0;
	}
}

} // namespace

extern "C" {

int pbdx_ensemble_create(pbdx_ensemble **out, const int *devices, uint32_t n)
{
	if (!out || !devices || !n) { set_error("pbdx_ensemble_create: null argument / no devices"); return PBDX_ERR_INVALID; }
	*out = nullptr;
	pbdx_ensemble *e = new (std::nothrow) pbdx_ensemble();
	if (!e) { set_error("out of memory"); return PBDX_ERR_ALLOC; }
	e->shards.resize(n);
	e->owner_pid = getpid();
	for (uint32_t i = 0; i < n; i++)
	{
		e->shards[i].device = devices[i];
		const int r = pbdx_timestep_create(&e->shards[i].ts, devices[i]);
		if (r != PBDX_OK) { pbdx_ensemble_destroy(e); return r; }
	}
	for (uint32_t i = 1; i < n; i++)
	{
		e->workers.emplace_back(new pbdx_ensemble::Worker());
		pbdx_ensemble::Worker *w = e->workers.back().get();
		w->th = std::thread(worker_main, w);
	}
	*out = e;
	return PBDX_OK;
}

void pbdx_ensemble_destroy(pbdx_ensemble *e)
{
	if (!e) return;
	for (std::unique_ptr<pbdx_ensemble::Worker> &w : e->workers)
	{
		if (e->owner_pid == getpid())
		{
			{ std::lock_guard<std::mutex> lk(w->mu); w->quit = true; w->cv.notify_one(); }
			if (w->th.joinable()) w->th.join();
		}
		else if (w->th.joinable()) w->th.detach();      // (forked child: the thread does not exist here)
	}
	drop_shards(e, false);
	delete e;
}

uint32_t pbdx_ensemble_num_shards(const pbdx_ensemble *e) { return e ? (uint32_t)e->shards.size() : 0u; }

int pbdx_ensemble_set_param(pbdx_ensemble *e, int id, int64_t value)
{
	if (!e) return PBDX_ERR_INVALID;
	for (pbdx_ensemble::Shard &s : e->shards) { const int r = pbdx_timestep_set_param(s.ts, id, value); if (r) return r; }
	return PBDX_OK;
}
int pbdx_ensemble_set_gravity(pbdx_ensemble *e, const float g[3])
{
	if (!e) return PBDX_ERR_INVALID;
	for (pbdx_ensemble::Shard &s : e->shards) { const int r = pbdx_timestep_set_gravity(s.ts, g); if (r) return r; }
	return PBDX_OK;
}
int pbdx_ensemble_set_time_step_size(pbdx_ensemble *e, float h)
{
	if (!e) return PBDX_ERR_INVALID;
	for (pbdx_ensemble::Shard &s : e->shards) { const int r = pbdx_timestep_set_time_step_size(s.ts, h); if (r) return r; }
	return PBDX_OK;
}

int pbdx_ensemble_set_model(pbdx_ensemble *e, const pbdx_model *m)
{
	if (!e || !m) return PBDX_ERR_INVALID;
	if (!m->size()) { set_error("pbdx_ensemble_set_model: empty model"); return PBDX_ERR_INVALID; }
	// (whatever happens below, the previous model is gone: a failure must not leave its identity behind with all blocks dropped -- ADVICE r5: the next
	// step then returned PBDX_OK having done nothing)
	e->have_model = false; e->whole_uid = 0;
	drop_shards(e, true);
	const uint64_t K = m->inst_count;
	const uint32_t world = (uint32_t)e->shards.size();
	for (uint32_t r = 0; r < world; r++)
	{
		pbdx_ensemble::Shard &s = e->shards[r];
		int rc = pbdx_ensemble_shard(K, world, r, &s.begin, &s.end);
		if (rc) { drop_shards(e, true); return rc; }
		if (s.begin == s.end) continue;                    // fewer instances than devices: this device stays idle
		std::string why;
		s.model = slice_model(m, s.begin, s.end, why);
		if (!s.model) { set_error("pbdx_ensemble_set_model: block [%llu, %llu): %s", (unsigned long long)s.begin, (unsigned long long)s.end, why.c_str()); drop_shards(e, true); return PBDX_ERR_INVALID; }
		rc = pbdx_timestep_invalidate(s.ts);
		if (rc) { drop_shards(e, true); return rc; }
	}
	e->whole_uid = m->uid; e->have_model = true;
	e->whole_topo = m->topology_version; e->whole_params = m->params_version; e->whole_state = m->state_version;
	return PBDX_OK;
}

int pbdx_ensemble_step(pbdx_ensemble *e, uint32_t num_steps)
{
	if (!e || !e->have_model) { set_error("pbdx_ensemble_step: no model (pbdx_ensemble_set_model)"); return PBDX_ERR_INVALID; }
	// the blocks are copies: the model may have been destroyed since (then there is nothing to compare with and the blocks step on); while it is alive an
	// edit that was not announced with another pbdx_ensemble_set_model is an error, as it is in pbdx_ensemble_gather
	if (const pbdx_model *w = find_model(e->whole_uid))
		if (w->topology_version != e->whole_topo || w->params_version != e->whole_params || w->state_version != e->whole_state)
		{ set_error("pbdx_ensemble_step: the model was edited since pbdx_ensemble_set_model (call it again: the blocks are copies)"); return PBDX_ERR_INVALID; }
	const auto t0 = std::chrono::steady_clock::now();
	const int r = for_all_shards(e, [&](size_t i) {
		pbdx_ensemble::Shard &s = e->shards[i];
		if (!s.model) return (int)PBDX_OK;
		const auto a = std::chrono::steady_clock::now();
		const int rc = pbdx_timestep_step_resident(s.ts, s.model, num_steps);      // enqueues on the shard's device and waits for ITS stream only
		s.last_step_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
		return rc;
	});
	e->last_step_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
	return r;
}

// every block's state back into the arrays of the whole model `m` (x, v, oldX, lastX): the ensemble's download
int pbdx_ensemble_gather(pbdx_ensemble *e, pbdx_model *m)
{
	if (!e || !m || !e->have_model || m->uid != e->whole_uid) { set_error("pbdx_ensemble_gather: not the model of pbdx_ensemble_set_model"); return PBDX_ERR_INVALID; }
	if (m->topology_version != e->whole_topo || m->params_version != e->whole_params || m->state_version != e->whole_state)
	{ set_error("pbdx_ensemble_gather: the model was edited since pbdx_ensemble_set_model (call it again: the blocks are copies)"); return PBDX_ERR_INVALID; }
	int r = for_all_shards(e, [&](size_t i) {
		pbdx_ensemble::Shard &s = e->shards[i];
		return s.model ? pbdx_timestep_sync_to_host(s.ts, s.model) : (int)PBDX_OK;
	});
	if (r) return r;
	const uint32_t np = m->inst_count > 1 ? m->inst_particles : m->size();
	for (const pbdx_ensemble::Shard &s : e->shards)
	{
		if (!s.model) continue;
		const size_t off = (size_t)3 * s.begin * np, cnt = (size_t)3 * (s.end - s.begin) * np;
		memcpy(m->x.data() + off, s.model->x.data(), cnt * sizeof(float));
		memcpy(m->v.data() + off, s.model->v.data(), cnt * sizeof(float));
		memcpy(m->old_x.data() + off, s.model->old_x.data(), cnt * sizeof(float));
		memcpy(m->last_x.data() + off, s.model->last_x.data(), cnt * sizeof(float));
	}
	// (the whole model's arrays now hold what the devices hold; they are not "edited": the blocks stay valid)
	return PBDX_OK;
}

int pbdx_ensemble_get_shard(const pbdx_ensemble *e, uint32_t shard, int *device, uint64_t *begin, uint64_t *end, double *last_step_ms)
{
	if (!e || shard >= e->shards.size()) { set_error("pbdx_ensemble_get_shard: no such shard"); return PBDX_ERR_INVALID; }
	const pbdx_ensemble::Shard &s = e->shards[shard];
	if (device) *device = s.device;
	if (begin) *begin = s.begin;
	if (end) *end = s.end;
	if (last_step_ms) *last_step_ms = s.last_step_ms;
	return PBDX_OK;
}
pbdx_timestep *pbdx_ensemble_timestep(pbdx_ensemble *e, uint32_t shard) { return (e && shard < e->shards.size()) ? e->shards[shard].ts : nullptr; }
pbdx_model *pbdx_ensemble_shard_model(pbdx_ensemble *e, uint32_t shard) { return (e && shard < e->shards.size()) ? e->shards[shard].model : nullptr; }
double pbdx_ensemble_last_step_ms(const pbdx_ensemble *e) { return e ? e->last_step_ms : 0.0; }

} // extern "C"
