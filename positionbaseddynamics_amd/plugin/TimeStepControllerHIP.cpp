#include "TimeStepControllerHIP.h"
#include "Simulation/Simulation.h"
#include "Simulation/TimeManager.h"
#include "Simulation/Constraints.h"
#include "Simulation/DistanceFieldCollisionDetection.h"
#include "Simulation/RigidBody.h"
#include "PositionBasedDynamics/TimeIntegration.h"
#include "Utils/Logger.h"
#include "Utils/Timing.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sched.h>
#include <pthread.h>
#include <unistd.h>
#include <new>
#include <algorithm>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <atomic>
#include <chrono>

using namespace PBD;

namespace
{
	// reference TYPE_IDs are run-time counters (Simulation/Constraints.cpp:17-49): map by identity
	int engineType(Constraint *c)
	{
		const int t = c->getTypeId();
		if (t == DistanceConstraint::TYPE_ID) return PBDX_DISTANCE;
		if (t == DistanceConstraint_XPBD::TYPE_ID) return PBDX_DISTANCE_XPBD;
		if (t == DihedralConstraint::TYPE_ID) return PBDX_DIHEDRAL;
		if (t == IsometricBendingConstraint::TYPE_ID) return PBDX_ISOMETRIC_BENDING;
		if (t == IsometricBendingConstraint_XPBD::TYPE_ID) return PBDX_ISOMETRIC_BENDING_XPBD;
		if (t == FEMTriangleConstraint::TYPE_ID) return PBDX_FEM_TRIANGLE;
		if (t == StrainTriangleConstraint::TYPE_ID) return PBDX_STRAIN_TRIANGLE;
		if (t == VolumeConstraint::TYPE_ID) return PBDX_VOLUME;
		if (t == VolumeConstraint_XPBD::TYPE_ID) return PBDX_VOLUME_XPBD;
		if (t == FEMTetConstraint::TYPE_ID) return PBDX_FEM_TET;
		if (t == XPBD_FEMTetConstraint::TYPE_ID) return PBDX_FEM_TET_XPBD;
		if (t == StrainTetConstraint::TYPE_ID) return PBDX_STRAIN_TET;
		if (t == ShapeMatchingConstraint::TYPE_ID && c->numberOfBodies() == 4) return PBDX_SHAPE_MATCHING;
		return -1;
	}

	// The parameter record of a constraint in the layout of include/pbdx.h, handed float by float to a sink: the same walk fills
	// the engine's parameter arrays (PushSink) and feeds the exact parameter scan (HashSink) -- one definition of "what the engine
	// was given", so the scan cannot miss a field the upload reads.
	template <class Sink, typename M> inline void visitColMajor(Sink &p, const M &m, int rows, int cols)
	{
		for (int c = 0; c < cols; c++) for (int r = 0; r < rows; r++) p((float)m(r, c));
	}
	template <class Sink> inline void visitParams(Sink &p, int type, Constraint *c)
	{
		switch (type)
		{
		case PBDX_DISTANCE: { auto *k = (DistanceConstraint*)c; p((float)k->m_restLength); p((float)k->m_stiffness); break; }
		case PBDX_DISTANCE_XPBD: { auto *k = (DistanceConstraint_XPBD*)c; p((float)k->m_restLength); p((float)k->m_stiffness); break; }
		case PBDX_DIHEDRAL: { auto *k = (DihedralConstraint*)c; p((float)k->m_restAngle); p((float)k->m_stiffness); break; }
		case PBDX_ISOMETRIC_BENDING: { auto *k = (IsometricBendingConstraint*)c; p((float)k->m_stiffness); visitColMajor(p, k->m_Q, 4, 4); break; }
		case PBDX_ISOMETRIC_BENDING_XPBD: { auto *k = (IsometricBendingConstraint_XPBD*)c; p((float)k->m_stiffness); visitColMajor(p, k->m_Q, 4, 4); break; }
		case PBDX_FEM_TRIANGLE: { auto *k = (FEMTriangleConstraint*)c; p((float)k->m_area); visitColMajor(p, k->m_invRestMat, 2, 2);
			p((float)k->m_xxStiffness); p((float)k->m_yyStiffness); p((float)k->m_xyStiffness);
			p((float)k->m_xyPoissonRatio); p((float)k->m_yxPoissonRatio); break; }
		case PBDX_STRAIN_TRIANGLE: { auto *k = (StrainTriangleConstraint*)c; visitColMajor(p, k->m_invRestMat, 2, 2);
			p((float)k->m_xxStiffness); p((float)k->m_yyStiffness); p((float)k->m_xyStiffness);
			p(k->m_normalizeStretch ? 1.0f : 0.0f); p(k->m_normalizeShear ? 1.0f : 0.0f); break; }
		case PBDX_VOLUME: { auto *k = (VolumeConstraint*)c; p((float)k->m_restVolume); p((float)k->m_stiffness); break; }
		case PBDX_VOLUME_XPBD: { auto *k = (VolumeConstraint_XPBD*)c; p((float)k->m_restVolume); p((float)k->m_stiffness); break; }
		case PBDX_FEM_TET: { auto *k = (FEMTetConstraint*)c; p((float)k->m_volume); visitColMajor(p, k->m_invRestMat, 3, 3);
			p((float)k->m_stiffness); p((float)k->m_poissonRatio); break; }
		case PBDX_FEM_TET_XPBD: { auto *k = (XPBD_FEMTetConstraint*)c; p((float)k->m_volume); visitColMajor(p, k->m_invRestMat, 3, 3);
			p((float)k->m_stiffness); p((float)k->m_poissonRatio); break; }
		case PBDX_STRAIN_TET: { auto *k = (StrainTetConstraint*)c; visitColMajor(p, k->m_invRestMat, 3, 3);
			p((float)k->m_stretchStiffness); p((float)k->m_shearStiffness);
			p(k->m_normalizeStretch ? 1.0f : 0.0f); p(k->m_normalizeShear ? 1.0f : 0.0f); break; }
		case PBDX_SHAPE_MATCHING: { auto *k = (ShapeMatchingConstraint*)c; p((float)k->m_stiffness);
			for (int j = 0; j < 3; j++) p((float)k->m_restCm[j]);
			for (int i = 0; i < 4; i++) for (int j = 0; j < 3; j++) p((float)k->m_x0[i][j]);
			for (int i = 0; i < 4; i++) p((float)k->m_w[i]);
			for (int i = 0; i < 4; i++) p((float)k->m_numClusters[i]);
			break; }
		default: break;
		}
	}
	struct PushSink { std::vector<float> &v; inline void operator()(float f) { v.push_back(f); } };
	inline void pushParams(std::vector<float> &p, int type, Constraint *c) { PushSink s = { p }; visitParams(s, type, c); }
	// Hash of parameter records: the XOR of pbdx_hash_word (include/pbdx.h: a bijective multiplicative mix) over the records' floats
	// taken two at a time, each pair mixed with (constraint index, position in the record).  The terms are independent of each other
	// (no serial chain through the multiplier's latency: the walk stays memory-bound), a single changed float always changes the
	// hash, and equal values in different constraints do not cancel.
	struct HashSink
	{
		uint64_t h; uint32_t idx, pend; bool half;
		inline void begin(uint32_t constraint) { idx = constraint << 5; half = false; }      // records hold at most 24 floats = 12 pairs
		inline void operator()(float f)
		{
			uint32_t u; memcpy(&u, &f, 4);
			if (!half) { pend = u; half = true; }
			else { h ^= pbdx_hash_word((uint64_t)pend | ((uint64_t)u << 32), idx++); half = false; }
		}
		inline void end() { if (half) { h ^= pbdx_hash_word((uint64_t)pend, idx++); half = false; } }
	};
#ifndef PBDX_SCAN_PREFETCH
#define PBDX_SCAN_PREFETCH 12     // objects ahead (scripts/dev/scan_bench.cpp)
#endif
	const size_t kParamScanBlock = 2048;          // constraints per 64-bit hash of the exact parameter scan

	// FNV-1a over raw bytes
	inline uint64_t fnv(uint64_t h, const void *p, size_t n)
	{
		const unsigned char *b = (const unsigned char *)p;
		for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
		return h;
	}
	// full-coverage block hashes of a packed host array (include/pbdx.h: pbdx_hash_block): every word of the array is hashed,
	// PBDX_HASH_BLOCK elements per 64-bit hash.  The pass is bandwidth-bound (one read of the array) and runs on a handful of
	// threads when the arrays are large (HostPool below).
	// The plug-in's own host passes (block hashes of the particle arrays, the exact parameter scan) run on a small pool of worker
	// threads of their OWN, not on the host application's OpenMP runtime: libgomp keeps one team of threads per host thread, and a
	// parallel region whose team size differs from the previous region's or from the host application's own setting costs milliseconds
	// (measured on the 256-CPU host of the MI355X box, profiles/HISTORY.md [8]: particle hashes 0.26 -> 8 ms, parameter scan 2.4 ->
	// 9-26 ms, from one step to the next).  The workers sleep on a condition variable between passes; the thread that calls step()
	// takes the first share itself.  Up to 64 threads (the parameter walk scales that far: 6 M constraints 7.9 / 4.0 / 2.4 ms at
	// 8 / 32 / 64; hosts commonly run their own loops with omp_set_num_threads(1), so the count does not follow OpenMP's).
	class HostPool
	{
	public:
		static HostPool &get() { static HostPool pool; return pool; }
		int threads() const { return m_threads; }
		// fn(begin, end) over a static partition of [0, n) into `parts` shares (parts <= threads())
		void run(size_t n, int parts, const std::function<void(size_t, size_t)> &fn)
		{
			if (parts > m_threads) parts = m_threads;
			if (parts <= 1 || n < 2) { fn(0, n); return; }
			// A forked child has this object but none of its threads, and its mutexes and condition variables in whatever state another thread held them
			// when the fork happened: it never touches them and runs its passes on the calling thread (no pthread_atfork handler, which would
			// outlive a dlclose of the plug-in).  The same for a job that calls run() itself (m_pass is not recursive).
			if (getpid() != m_pid || t_inPass) { fn(0, n); return; }
			struct InPass { InPass() { t_inPass = true; } ~InPass() { t_inPass = false; } } inPass;
			// one pass at a time: the pool is shared by every controller of the process, and two controllers may be stepped from two host threads
			// (the single m_fn / m_pending state raced)
			std::lock_guard<std::mutex> pass(m_pass);
			start();
			{
				std::lock_guard<std::mutex> lk(m_mutex);
				m_fn = &fn; m_n = n; m_parts = parts; m_pending = parts - 1; m_generation++;
			}
			m_wake.notify_all();
			fn(0, n / (size_t)parts);                                // share 0 on the calling thread
			std::unique_lock<std::mutex> lk(m_mutex);
			m_done.wait(lk, [this] { return m_pending == 0; });
			m_fn = nullptr;
		}
	private:
		HostPool() : m_threads(1), m_fn(nullptr), m_n(0), m_parts(0), m_pending(0), m_generation(0), m_stop(false)
		{
			// the CPUs this process may run on (affinity mask: cgroup / taskset limits), not the machine's
			unsigned int hw = 0;
			cpu_set_t set;
			if (sched_getaffinity(0, sizeof(set), &set) == 0) hw = (unsigned int)CPU_COUNT(&set);
			if (!hw) hw = std::thread::hardware_concurrency();
			m_threads = hw ? (int)hw : 1;
			// (128 threads on the 256-CPU host were tried in round 5: the scan does not get faster and the particle hashes get three times slower, profiles/HISTORY.md [9])
			if (m_threads > 64) m_threads = 64;
			if (const char *e = getenv("PBDX_PLUGIN_HASH_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 256) m_threads = v; }      // developer aid
			m_pid = getpid();
		}
		~HostPool()
		{
			{ std::lock_guard<std::mutex> lk(m_mutex); m_stop = true; }
			m_wake.notify_all();
			for (std::thread &t : m_workers) if (t.joinable()) t.join();
		}
		void start()
		{
			if (!m_workers.empty() || m_threads <= 1) return;
			for (int k = 1; k < m_threads; k++) m_workers.emplace_back([this, k] { work(k); });
		}
		void work(int k)
		{
			uint64_t seen = 0;
			for (;;)
			{
				const std::function<void(size_t, size_t)> *fn; size_t n; int parts;
				{
					std::unique_lock<std::mutex> lk(m_mutex);
					m_wake.wait(lk, [&] { return m_stop || m_generation != seen; });
					if (m_stop) return;
					seen = m_generation; fn = m_fn; n = m_n; parts = m_parts;
				}
				if (k < parts)
				{
					t_inPass = true;
					(*fn)(n * (size_t)k / (size_t)parts, n * (size_t)(k + 1) / (size_t)parts);
					t_inPass = false;
					std::lock_guard<std::mutex> lk(m_mutex);
					if (--m_pending == 0) m_done.notify_one();
				}
			}
		}
		int m_threads;
		pid_t m_pid;
		static thread_local bool t_inPass;
		std::vector<std::thread> m_workers;
		std::mutex m_mutex, m_pass;
		std::condition_variable m_wake, m_done;
		const std::function<void(size_t, size_t)> *m_fn;
		size_t m_n; int m_parts, m_pending; uint64_t m_generation; bool m_stop;
	};
	thread_local bool HostPool::t_inPass = false;
	struct HashJob { const void *base; uint32_t n, elemBytes; std::vector<uint64_t> *out; };
	void hashBlocks(HashJob *jobs, int numJobs)
	{
		size_t bytes = 0;
		int total = 0;
		int first[8];
		for (int j = 0; j < numJobs; j++)
		{
			first[j] = total;
			const int nb = (int)pbdx_hash_num_blocks(jobs[j].n);
			jobs[j].out->resize((size_t)nb);
			total += nb;
			bytes += (size_t)jobs[j].n * jobs[j].elemBytes;
		}
		first[numJobs] = total;
		// ONE pass over the blocks of all arrays
		HostPool::get().run((size_t)total, bytes >= ((size_t)1 << 20) ? HostPool::get().threads() : 1, [&](size_t q0, size_t q1)
		{
			for (size_t q = q0; q < q1; q++)
			{
				int j = 0;
				while ((int)q >= first[j + 1]) j++;
				(*jobs[j].out)[q - (size_t)first[j]] = pbdx_hash_block(jobs[j].base, jobs[j].n, jobs[j].elemBytes, (uint32_t)(q - (size_t)first[j]));
			}
		});
	}
	// element ranges (first, count pairs; adjacent blocks merged) covered by the blocks whose hash differs
	void changedRanges(const std::vector<uint64_t> &was, const std::vector<uint64_t> &now, uint32_t n, std::vector<uint32_t> &ranges)
	{
		ranges.clear();
		const size_t nb = now.size();
		if (was.size() != nb) { if (n) { ranges.push_back(0); ranges.push_back(n); } return; }
		for (size_t b = 0; b < nb; b++)
		{
			if (was[b] == now[b]) continue;
			const uint32_t first = (uint32_t)(b * PBDX_HASH_BLOCK);
			const uint32_t count = n - first < PBDX_HASH_BLOCK ? n - first : PBDX_HASH_BLOCK;
			if (!ranges.empty() && ranges[ranges.size() - 2] + ranges.back() == first) ranges.back() += count;
			else { ranges.push_back(first); ranges.push_back(count); }
		}
		// many scattered edits: one range over everything costs less than hundreds of small copies
		if (ranges.size() > 2 * 256) { ranges.clear(); ranges.push_back(0); ranges.push_back(n); }
	}
	inline double nowMs() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
	struct Lap { double *acc, t0; Lap(double *a) : acc(a), t0(nowMs()) {} ~Lap() { *acc += nowMs() - t0; } };
	static_assert(sizeof(Vector3r) == 3 * sizeof(Real), "ParticleData's std::vector<Vector3r> must be a packed Real[3] array (Common/Common.h:31, Eigen::DontAlign)");
}

TimeStepControllerHIP::TimeStepControllerHIP(int device) :
	TimeStepController(), m_solver(nullptr), m_device(device), m_scheduleValid(false),
	m_numConstraints(0), m_numParticles(0), m_gpuSteps(0), m_fallbackSteps(0), m_failedSteps(0), m_paramRefreshes(0), m_scheduleBuilds(0), m_uploads(0),
	m_allowFallback(false), m_deviceAhead(false), m_hostDirty(false), m_imageValid(false), m_paramsDirty(false),
	m_supportedFor(nullptr), m_supportedConstraints(0), m_supportedBodies(0), m_supportedObjects(0), m_supported(false), m_accelValid(false), m_tetSignature(0)
{
	m_accelGravity[0] = m_accelGravity[1] = m_accelGravity[2] = 0;
	m_fullParameterScan = true; m_partialUploads = 0; m_mixed = false; m_mixedGroupsLast = 0; m_dynamicBodies = false;
	m_speculate = getenv("PBDX_PLUGIN_NO_SPECULATION") == NULL; m_speculativeSteps = 0; m_repeatedSteps = 0; m_failRepeatForTest = false;
	m_rawH = 0.0f; m_rawG[0] = m_rawG[1] = m_rawG[2] = 0.0f;
	for (int k = 0; k < 6; k++) m_ms[k] = 0.0;
	m_deviceMs = 0.0;
	if (pbdx_solver_create(&m_solver, device) != PBDX_OK)
	{
		LOG_ERR << "TimeStepControllerHIP: " << pbdx_last_error() << " -- every step() will fail (no CPU path)";
		m_solver = nullptr;
	}
	// ParticleData's arrays are handed to the engine in place; PBDX_OPT_PIN_HOST moves them through a page-locked mirror the engine owns, so that
	// the per-step round trip of step() runs at the PCIe rate without the reference's heap being registered with the GPU
	// (setPinHostArrays(false) switches that off)
	if (m_solver) pbdx_solver_set_option(m_solver, PBDX_OPT_PIN_HOST, 1);
}

// A step the engine cannot run: loud error, model untouched -- unless the host opted in to the
// reference's own CPU path.
void TimeStepControllerHIP::refuse(SimulationModel &model, const char *why)
{
	if (m_allowFallback)
	{
		LOG_WARN << "TimeStepControllerHIP: " << why << " -- running this step on the reference CPU path (opted in)";
		m_fallbackSteps++;
		if (m_deviceAhead) syncToHost(model);               // the CPU path continues from the newest state
		TimeStepController::step(model);
		m_hostDirty = true;
		return;
	}
	LOG_ERR << "TimeStepControllerHIP: " << why << " -- step NOT executed (call setAllowReferenceFallback(true) to use the CPU TimeStepController)";
	fprintf(stderr, "TimeStepControllerHIP: %s -- step NOT executed\n", why);
	m_failedSteps++;
}

TimeStepControllerHIP::~TimeStepControllerHIP()
{
	pbdx_solver_destroy(m_solver);
}

void TimeStepControllerHIP::reset()
{
	TimeStepController::reset();
	m_scheduleValid = false;
	// Simulation::reset resets the model on the host (SimulationModel::reset): the host is authoritative again
	m_deviceAhead = false;
	m_hostDirty = true;
	m_accelValid = false;
	// SimulationModel::reset -> resetContacts: the contact list of the last detection goes with it
	if (m_solver && m_tetSignature) { pbdx_solver_set_tet_colliders(m_solver, 0, NULL, 0.0f); m_tetSignature = 0; }
}

// analytic distance fields the engine evaluates (pbdx_contact.h); parameters as the reference's collision objects store them
static bool analyticShape(CollisionDetection::CollisionObject *co, int *shape, float *params)
{
	typedef DistanceFieldCollisionDetection D;
	const int t = co->getTypeId();
	int sh; float p[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
	if (t == D::DistanceFieldCollisionBox::TYPE_ID) { sh = PBDX_SHAPE_BOX; for (int k = 0; k < 3; k++) p[k] = (float)((D::DistanceFieldCollisionBox*)co)->m_box[k]; }
	else if (t == D::DistanceFieldCollisionSphere::TYPE_ID) { sh = PBDX_SHAPE_SPHERE; p[0] = (float)((D::DistanceFieldCollisionSphere*)co)->m_radius; }
	else if (t == D::DistanceFieldCollisionTorus::TYPE_ID) { sh = PBDX_SHAPE_TORUS; p[0] = (float)((D::DistanceFieldCollisionTorus*)co)->m_radii[0]; p[1] = (float)((D::DistanceFieldCollisionTorus*)co)->m_radii[1]; }
	else if (t == D::DistanceFieldCollisionCylinder::TYPE_ID) { sh = PBDX_SHAPE_CYLINDER; p[0] = (float)((D::DistanceFieldCollisionCylinder*)co)->m_dim[0]; p[1] = (float)((D::DistanceFieldCollisionCylinder*)co)->m_dim[1]; }
	else if (t == D::DistanceFieldCollisionHollowSphere::TYPE_ID) { sh = PBDX_SHAPE_HOLLOW_SPHERE; p[0] = (float)((D::DistanceFieldCollisionHollowSphere*)co)->m_radius; p[1] = (float)((D::DistanceFieldCollisionHollowSphere*)co)->m_thickness; }
	else if (t == D::DistanceFieldCollisionHollowBox::TYPE_ID) { sh = PBDX_SHAPE_HOLLOW_BOX; for (int k = 0; k < 3; k++) p[k] = (float)((D::DistanceFieldCollisionHollowBox*)co)->m_box[k]; p[3] = (float)((D::DistanceFieldCollisionHollowBox*)co)->m_thickness; }
	else return false;
	if (shape) *shape = sh;
	if (params) for (int k = 0; k < 4; k++) params[k] = p[k];
	return true;
}

// structure of one of the reference's bounding-sphere hierarchies (kdTree.h: entity list, nodes, spheres), which the REFERENCE
// built when the collision object was registered.  The node count has no accessor: nodes are numbered in creation order, so it is
// the largest index reachable from the root + 1.
struct HierarchyCopy { std::vector<uint32_t> lst; std::vector<int32_t> nodes; std::vector<float> hulls; };
template <class BVH> static void copyHierarchy(const BVH &bvh, unsigned int numEntities, HierarchyCopy &out, pbdx_bvh &rec)
{
	std::vector<unsigned int> stack(1, 0u);
	unsigned int maxIdx = 0;
	while (!stack.empty())
	{
		const unsigned int n = stack.back(); stack.pop_back();
		if (n > maxIdx) maxIdx = n;
		if (!bvh.node(n).is_leaf()) { stack.push_back((unsigned int)bvh.node(n).children[0]); stack.push_back((unsigned int)bvh.node(n).children[1]); }
	}
	const unsigned int nNodes = maxIdx + 1;
	out.lst.resize(numEntities); out.nodes.resize((size_t)4 * nNodes); out.hulls.resize((size_t)4 * nNodes);
	for (unsigned int i = 0; i < numEntities; i++) out.lst[i] = bvh.entity(i);
	for (unsigned int i = 0; i < nNodes; i++)
	{
		out.nodes[4 * i] = bvh.node(i).children[0]; out.nodes[4 * i + 1] = bvh.node(i).children[1];
		out.nodes[4 * i + 2] = (int32_t)bvh.node(i).begin; out.nodes[4 * i + 3] = (int32_t)bvh.node(i).n;
		for (int k = 0; k < 3; k++) out.hulls[4 * i + k] = (float)bvh.hull(i).x()[k];
		out.hulls[4 * i + 3] = (float)bvh.hull(i).r();
	}
	rec.num_nodes = nNodes; rec.num_entities = numEntities; rec.entities = out.lst.data(); rec.nodes = out.nodes.data(); rec.hulls = out.hulls.data();
}

// The scan over all constraints (one virtual call each) is repeated only when the model's make-up changed.
bool TimeStepControllerHIP::supported(SimulationModel &model)
{
	if (!m_solver) return false;
	const size_t nObjects = m_collisionDetection != NULL ? m_collisionDetection->getCollisionObjects().size() : 0;
	if (m_supportedFor == (const void *)&model && m_supportedConstraints == model.getConstraints().size() &&
		m_supportedBodies == model.getRigidBodies().size() && m_supportedObjects == nObjects && model.m_groupsInitialized && m_scheduleValid)
	{
		// coefficients are host-mutable between steps: the friction rule for deformable-deformable contacts is checked every step
		// (O(number of tet collision objects))
		if (m_supported && m_collisionDetection != NULL)
		{
			unsigned int tetObjects = 0; bool friction = false;
			for (CollisionDetection::CollisionObject *co : m_collisionDetection->getCollisionObjects())
				if (co->m_bodyType == CollisionDetection::CollisionObject::TetModelCollisionObjectType)
				{
					tetObjects++;
					if (model.getTetModels()[co->m_bodyIndex]->getFrictionCoeff() != 0.0) friction = true;
				}
			if (tetObjects > 1 && friction) return false;
		}
		return m_supported;
	}
	m_supportedFor = (const void *)&model; m_supportedConstraints = model.getConstraints().size();
	m_supportedBodies = model.getRigidBodies().size(); m_supportedObjects = nObjects;
	m_supported = false;
	if (model.getOrientations().size() != 0) return false;
	// rigid bodies: as colliders of a distance-field collision detection.  Static ones (mass 0) always; bodies of finite mass as impulse sinks of the
	// particle contacts (include/pbdx.h) -- their own dynamics stay on the host (integrateBodies) -- as long as no contact BETWEEN rigid bodies can
	// arise (RigidBodyContactConstraint is outside the path: a dynamic body next to another rigid collision object is only taken when neither tests its
	// mesh against the other, DistanceFieldCollisionDetection.cpp:112-124) and no tet model collides with tet models at the same time
	m_dynamicBodies = false;
	for (RigidBody *rb : model.getRigidBodies())
		if (rb->getMass() != 0.0) m_dynamicBodies = true;
	if (!model.getRigidBodies().empty() && m_collisionDetection == NULL) return false;
	if (m_collisionDetection != NULL && dynamic_cast<DistanceFieldCollisionDetection*>(m_collisionDetection) == NULL) return false;
	if (m_dynamicBodies)
	{
		typedef DistanceFieldCollisionDetection D;
		unsigned int rigidObjects = 0, rigidTesting = 0, tetObjects = 0;
		for (CollisionDetection::CollisionObject *co : m_collisionDetection->getCollisionObjects())
		{
			if (co->m_bodyType == CollisionDetection::CollisionObject::RigidBodyCollisionObjectType)
			{
				rigidObjects++;
				if (((D::DistanceFieldCollisionObject*)co)->m_testMesh) rigidTesting++;
			}
			else if (co->m_bodyType == CollisionDetection::CollisionObject::TetModelCollisionObjectType) tetObjects++;
		}
		if (rigidObjects > 1 && rigidTesting != 0) return false;
		if (tetObjects > 1) return false;
		if (rigidObjects > 32) return false;
	}
	if (m_collisionDetection != NULL)
	{
		typedef DistanceFieldCollisionDetection D;
		for (CollisionDetection::CollisionObject *co : m_collisionDetection->getCollisionObjects())
		{
			const int t = co->getTypeId();
			if (co->m_bodyType == CollisionDetection::CollisionObject::RigidBodyCollisionObjectType)
			{
				if (t != D::DistanceFieldCollisionBox::TYPE_ID && t != D::DistanceFieldCollisionSphere::TYPE_ID && t != D::DistanceFieldCollisionTorus::TYPE_ID &&
					t != D::DistanceFieldCollisionCylinder::TYPE_ID && t != D::DistanceFieldCollisionHollowSphere::TYPE_ID && t != D::DistanceFieldCollisionHollowBox::TYPE_ID)
					return false;                             // e.g. cubic SDF (Discregrid) colliders
			}
			else if (co->m_bodyType == CollisionDetection::CollisionObject::TetModelCollisionObjectType)
			{
				if (t != D::DistanceFieldCollisionObjectWithoutGeometry::TYPE_ID && !analyticShape(co, NULL, NULL)) return false;      // e.g. cubic SDF (Discregrid)
			}
			else if (t != D::DistanceFieldCollisionObjectWithoutGeometry::TYPE_ID)
				return false;
		}
		// more than one tet model registered => every ordered pair of them is tested for deformable-deformable contacts
		// (ParticleTetContactConstraint, DistanceFieldCollisionDetection.cpp:160-177).  The engine runs them when every tet object
		// carries an analytic distance field (the reference walks the second object's tet hierarchy, which exists only then) and the
		// friction of every pair is zero (the reference's friction impulse for these contacts reads an unset multiplier).
		unsigned int tetObjects = 0, tetWithout = 0;
		bool friction = false;
		for (CollisionDetection::CollisionObject *co : m_collisionDetection->getCollisionObjects())
			if (co->m_bodyType == CollisionDetection::CollisionObject::TetModelCollisionObjectType)
			{
				tetObjects++;
				if (co->getTypeId() == D::DistanceFieldCollisionObjectWithoutGeometry::TYPE_ID) tetWithout++;
				if (model.getTetModels()[co->m_bodyIndex]->getFrictionCoeff() != 0.0) friction = true;
			}
		if (tetObjects > 1 && (tetWithout != 0 || friction)) return false;
	}
	// Constraint classes the engine does not know (PositionBasedGenericConstraints.h templates, user subclasses): a MIXED model -- the
	// known (group, type) buckets run on the GPU, the others through the reference's own virtual solvePositionConstraint on the host
	// inside the same colour groups (runMixedSteps).  Only for pure particle models: an unknown class may read anything, and what the
	// plug-in keeps current on the host between the groups is ParticleData's positions (+ masses).
	// (a walk over every heap object of the model: by the worker pool -- at 6 M constraints a single thread needs 0.2-0.3 s for it, and the first step made
	// three such walks one after the other: this one, the bodies for the colouring and the schedule, profiles/HISTORY.md [10])
	std::atomic<size_t> unknownCount(0);
	{
		SimulationModel::ConstraintVector &cs = model.getConstraints();
		HostPool &pool = HostPool::get();
		pool.run(cs.size(), cs.size() > 65536 ? pool.threads() : 1, [&](size_t b, size_t e) {
			size_t u = 0;
			for (size_t i = b; i < e; i++) if (engineType(cs[i]) < 0) u++;                   // e.g. GenericConstraints, joints, rods
			if (u) unknownCount += u;
		});
	}
	const size_t unknown = unknownCount.load();
	if (unknown && (!model.getRigidBodies().empty() || nObjects != 0)) return false;
	m_mixed = unknown != 0;
	m_supported = true;
	return true;
}

// Colliders = the rigid-body collision objects of the reference's DistanceFieldCollisionDetection, with
// the transformation the reference keeps per rigid body (RigidBody::getTransformationR/V1/V2); collision
// ranges = the triangle / tet models registered with testMesh (DistanceFieldCollisionDetection.cpp:124-154).
bool TimeStepControllerHIP::uploadColliders(SimulationModel &model)
{
	std::vector<pbdx_collider> cols;
	std::vector<pbdx_collision_range> ranges;
	std::vector<pbdx_collider_dynamics> dyns;
	std::vector<uint32_t> rangeObject, rank;
	m_colliderBody.clear();
	float tolerance = 0.01f;
	if (m_collisionDetection != NULL)
	{
		typedef DistanceFieldCollisionDetection D;
		tolerance = (float)m_collisionDetection->getTolerance();
		uint32_t objectIndex = 0;
		for (CollisionDetection::CollisionObject *co : m_collisionDetection->getCollisionObjects())
		{
			const int t = co->getTypeId();
			if (co->m_bodyType == CollisionDetection::CollisionObject::RigidBodyCollisionObjectType)
			{
				pbdx_collider c;
				memset(&c, 0, sizeof(c));
				D::DistanceFieldCollisionObject *dco = (D::DistanceFieldCollisionObject*)co;
				c.invert = dco->m_invertSDF < 0 ? 1 : 0;
				analyticShape(co, &c.shape, c.params);
				RigidBody *rb = model.getRigidBodies()[co->m_bodyIndex];
				const Matrix3r &R = rb->getTransformationR();
				for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) c.R[3 * r + k] = (float)R(r, k);
				for (int k = 0; k < 3; k++)
				{
					c.com[k] = (float)rb->getPosition()[k];
					c.v1[k] = (float)rb->getTransformationV1()[k];
					c.v2[k] = (float)rb->getTransformationV2()[k];
					c.body_v[k] = (float)rb->getVelocity()[k];
					c.body_omega[k] = (float)rb->getAngularVelocity()[k];
				}
				c.restitution = (float)rb->getRestitutionCoeff();
				c.friction = (float)rb->getFrictionCoeff();
				c.body_index = co->m_bodyIndex;
				cols.push_back(c);
				pbdx_collider_dynamics d;
				memset(&d, 0, sizeof(d));
				d.inv_mass = (float)rb->getInvMass();
				const Matrix3r &Ji = rb->getInertiaTensorInverseW();
				for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) d.inertia_inv_w[3 * r + k] = (float)Ji(r, k);
				d.object_index = objectIndex;
				dyns.push_back(d);
				m_colliderBody.push_back(co->m_bodyIndex);
			}
			else if (((D::DistanceFieldCollisionObject*)co)->m_testMesh)
			{
				pbdx_collision_range r;
				if (co->m_bodyType == CollisionDetection::CollisionObject::TriangleModelCollisionObjectType)
				{
					TriangleModel *tm = model.getTriangleModels()[co->m_bodyIndex];
					r.first = tm->getIndexOffset(); r.count = tm->getParticleMesh().numVertices();
					r.restitution = (float)tm->getRestitutionCoeff(); r.friction = (float)tm->getFrictionCoeff();
				}
				else
				{
					TetModel *tm = model.getTetModels()[co->m_bodyIndex];
					r.first = tm->getIndexOffset(); r.count = tm->getParticleMesh().numVertices();
					r.restitution = (float)tm->getRestitutionCoeff(); r.friction = (float)tm->getFrictionCoeff();
				}
				ranges.push_back(r);
				rangeObject.push_back(objectIndex);
				if (m_dynamicBodies)
				{
					// a particle's place in the reference's contact order: its position in the entity list of the object's point hierarchy, which the
					// reference walks depth first, left to right (collisionDetectionRBSolid, kdTree.inl:84-105)
					if (rank.empty()) rank.assign(model.getParticles().size(), 0u);
					const PointCloudBSH &bvh = ((D::DistanceFieldCollisionObject*)co)->m_bvh;
					for (unsigned int q = 0; q < r.count; q++) rank[r.first + bvh.entity(q)] = q;
				}
			}
			objectIndex++;
		}
	}
	if (pbdx_solver_set_colliders(m_solver, (uint32_t)cols.size(), cols.data()) != PBDX_OK) return false;
	if (pbdx_solver_set_collision_ranges(m_solver, (uint32_t)ranges.size(), ranges.data()) != PBDX_OK) return false;
	if (pbdx_solver_set_collider_dynamics(m_solver, m_dynamicBodies ? (uint32_t)dyns.size() : 0u, dyns.data()) != PBDX_OK) return false;
	if (m_dynamicBodies && !ranges.empty())
	{
		if (rank.size() != model.getParticles().size()) rank.assign(model.getParticles().size(), 0u);
		// (the hierarchies' entity lists are fixed when the objects are registered: uploaded again only if something about them changed)
		if (rank != m_contactRank || rangeObject != m_contactRangeObject)
		{
			if (pbdx_solver_set_contact_order(m_solver, (uint32_t)rangeObject.size(), rangeObject.data(), (uint32_t)rank.size(), rank.data()) != PBDX_OK) return false;
			m_contactRank = rank; m_contactRangeObject = rangeObject;
		}
	}
	if (!uploadTetColliders(model, tolerance)) return false;
	return pbdx_solver_set_contact_params(m_solver, tolerance, (float)model.getContactStiffnessParticleRigidBody(), m_maxIterationsV) == PBDX_OK;
}

// Deformable vs deformable contacts: the tet-model collision objects with their distance field in the rest frame and the three
// hierarchies the reference built for them (DistanceFieldCollisionDetection.cpp:485-520 addCollisionObject + initTetBVH).  Uploaded
// once per set of objects: the engine keeps the contact list of the last detection next to them (it is what the NEXT step's
// position solves read), so a re-upload is also what empties that list (reset()).
bool TimeStepControllerHIP::uploadTetColliders(SimulationModel &model, float tolerance)
{
	typedef DistanceFieldCollisionDetection D;
	std::vector<D::DistanceFieldCollisionObject*> objs;
	if (m_collisionDetection != NULL)
		for (CollisionDetection::CollisionObject *co : m_collisionDetection->getCollisionObjects())
			if (co->m_bodyType == CollisionDetection::CollisionObject::TetModelCollisionObjectType && analyticShape(co, NULL, NULL))
				objs.push_back((D::DistanceFieldCollisionObject*)co);
	if (objs.size() < 2) objs.clear();                       // a single solid has nobody to collide with (no self collisions, :43)
	uint64_t sig = 0;
	if (!objs.empty())
	{
		const unsigned int n = model.getParticles().size();
		sig = fnv(fnv(fnv(1469598103934665603ull, objs.data(), objs.size() * sizeof(objs[0])), &tolerance, sizeof(tolerance)), &n, sizeof(n));
		// everything the engine's copy of a collider was made from and the host can change at run time: distance field, flags,
		// coefficients, the model's rest frame, the extent of the three hierarchies (a fresh initTetBVH)
		for (D::DistanceFieldCollisionObject *co : objs)
		{
			TetModel *tm = model.getTetModels()[co->m_bodyIndex];
			int shape = 0; float params[4];
			analyticShape(co, &shape, params);
			const int flags[3] = { shape, co->m_invertSDF < 0 ? 1 : 0, co->m_testMesh ? 1 : 0 };
			const Real coeff[2] = { tm->getFrictionCoeff(), tm->getRestitutionCoeff() };
			const unsigned int extent[8] = { co->m_bodyIndex, tm->getIndexOffset(), tm->getParticleMesh().numVertices(), tm->getParticleMesh().numTets(),
				co->m_bvh.node(0).n, co->m_bvhTets.node(0).n, co->m_bvhTets0.node(0).n, (unsigned int)co->m_bvh.node(0).children[0] };
			sig = fnv(fnv(fnv(fnv(sig, flags, sizeof(flags)), params, sizeof(params)), coeff, sizeof(coeff)), extent, sizeof(extent));
			sig = fnv(fnv(sig, &tm->getInitialX()[0], 3 * sizeof(Real)), tm->getInitialR().data(), 9 * sizeof(Real));
		}
		if (!sig) sig = 1;
	}
	if (sig == m_tetSignature) return true;
	m_tetSignature = 0;
	if (objs.empty()) return pbdx_solver_set_tet_colliders(m_solver, 0, NULL, tolerance) == PBDX_OK;
	ParticleData &pd = model.getParticles();
	const unsigned int n = pd.size();
#ifdef USE_DOUBLE
	std::vector<float> x0((size_t)3 * n);
	for (unsigned int i = 0; i < n; i++) for (int k = 0; k < 3; k++) x0[3 * i + k] = (float)pd.getPosition0(i)[k];
	if (pbdx_solver_set_rest_positions(m_solver, n, x0.data()) != PBDX_OK) return false;
#else
	if (pbdx_solver_set_rest_positions(m_solver, n, &pd.getPosition0(0)[0]) != PBDX_OK) return false;
#endif
	std::vector<pbdx_tet_collider> recs(objs.size());
	std::vector<HierarchyCopy> copies(3 * objs.size());
	std::vector<std::vector<uint32_t> > tets(objs.size());
	for (size_t q = 0; q < objs.size(); q++)
	{
		D::DistanceFieldCollisionObject *co = objs[q];
		TetModel *tm = model.getTetModels()[co->m_bodyIndex];
		const Utilities::IndexedTetMesh &mesh = tm->getParticleMesh();
		pbdx_tet_collider &c = recs[q];
		memset(&c, 0, sizeof(c));
		analyticShape(co, &c.shape, c.params);
		c.invert = co->m_invertSDF < 0 ? 1 : 0;
		c.first_particle = tm->getIndexOffset(); c.num_vertices = mesh.numVertices(); c.num_tets = mesh.numTets();
		tets[q].assign(mesh.getTets().begin(), mesh.getTets().end());
		c.tets = tets[q].data();
		for (int k = 0; k < 3; k++) c.initial_x[k] = (float)tm->getInitialX()[k];
		for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) c.initial_R[3 * r + k] = (float)tm->getInitialR()(r, k);
		c.restitution = (float)tm->getRestitutionCoeff(); c.friction = (float)tm->getFrictionCoeff();
		c.test_mesh = co->m_testMesh ? 1 : 0; c.body_index = co->m_bodyIndex;
		copyHierarchy(co->m_bvh, c.num_vertices, copies[3 * q], c.points);
		copyHierarchy(co->m_bvhTets, c.num_tets, copies[3 * q + 1], c.tets_bvh);
		copyHierarchy(co->m_bvhTets0, c.num_tets, copies[3 * q + 2], c.tets_rest);
	}
	if (pbdx_solver_set_tet_colliders(m_solver, (uint32_t)recs.size(), recs.data(), tolerance) != PBDX_OK) return false;
	m_tetSignature = sig;
	return true;
}

// ParticleData's arrays go to the engine as they are: std::vector<Vector3r> is a packed Real[3] array
// (ParticleData.h:91-100), so &getPosition(0)[0] is what pbdx_solver_set_particles expects in a float host; a double
// host hands the same pointers to the _f64 entry point and the conversion happens on the device.  Only the inverse
// masses have no reference accessor (ParticleData.h:248 returns by value): they are gathered into a scratch array.
bool TimeStepControllerHIP::uploadParticles(SimulationModel &model)
{
	ParticleData &pd = model.getParticles();
	const unsigned int n = pd.size();
	if (n != m_numParticles) { m_scheduleValid = false; m_contactRank.clear(); m_contactRangeObject.clear(); }      // the engine drops its schedule (and the contact order) with the old particle image
	m_numParticles = n;
	m_uploads++;
	m_accelValid = false;
	// a full upload always gathers the inverse masses afresh (one linear pass, negligible next to the transfer): the host is
	// authoritative and ParticleData::setMass may have pinned / released any particle since the last upload
	m_invMass.resize(n);
	for (unsigned int i = 0; i < n; i++) m_invMass[i] = pd.getInvMass(i);
	int r;
#ifdef USE_DOUBLE
	r = n ? pbdx_solver_set_particles_f64(m_solver, n, &pd.getPosition(0)[0], &pd.getVelocity(0)[0], &pd.getOldPosition(0)[0], &pd.getLastPosition(0)[0],
		&pd.getMass(0), m_invMass.data()) : PBDX_OK;
#else
	r = n ? pbdx_solver_set_particles(m_solver, n, &pd.getPosition(0)[0], &pd.getVelocity(0)[0], &pd.getOldPosition(0)[0], &pd.getLastPosition(0)[0],
		&pd.getMass(0), m_invMass.data()) : PBDX_OK;
#endif
	if (r != PBDX_OK) return false;
	m_imageValid = true;
	m_deviceAhead = false;
	m_hostDirty = false;
	hashHostState(model, m_blockHash);
	return true;
}

// Host is current (nothing newer on the device): only the blocks the host wrote since the last upload / download go to the
// device.  Changed masses re-derive the inverse masses of their blocks (ParticleData::setMass, ParticleData.h:239-246).
bool TimeStepControllerHIP::uploadChanges(SimulationModel &model, std::vector<uint64_t> now[5])
{
	ParticleData &pd = model.getParticles();
	const unsigned int n = pd.size();
	const Real *base[5] = { &pd.getPosition(0)[0], &pd.getVelocity(0)[0], &pd.getOldPosition(0)[0], &pd.getLastPosition(0)[0], &pd.getMass(0) };
	std::vector<uint32_t> ranges;
	bool any = false;
	for (int k = 0; k < 5; k++)
	{
		changedRanges(m_blockHash[k], now[k], n, ranges);
		if (ranges.empty()) continue;
		any = true;
		const uint32_t nr = (uint32_t)(ranges.size() / 2);
#ifdef USE_DOUBLE
		if (pbdx_solver_update_particle_ranges_f64(m_solver, k, base[k], nr, ranges.data()) != PBDX_OK) return false;
#else
		if (pbdx_solver_update_particle_ranges(m_solver, k, base[k], nr, ranges.data()) != PBDX_OK) return false;
#endif
		if (k == 4)
		{
			if (m_invMass.size() != n) { m_invMass.resize(n); ranges.assign(1, 0u); ranges.push_back(n); }
			for (size_t r = 0; r + 1 < ranges.size(); r += 2)
				for (uint32_t i = ranges[r]; i < ranges[r] + ranges[r + 1]; i++) m_invMass[i] = pd.getInvMass(i);
#ifdef USE_DOUBLE
			if (pbdx_solver_update_particle_ranges_f64(m_solver, PBDX_ARRAY_INV_MASS, m_invMass.data(), (uint32_t)(ranges.size() / 2), ranges.data()) != PBDX_OK) return false;
#else
			if (pbdx_solver_update_particle_ranges(m_solver, PBDX_ARRAY_INV_MASS, m_invMass.data(), (uint32_t)(ranges.size() / 2), ranges.data()) != PBDX_OK) return false;
#endif
			m_accelValid = false;
		}
		m_blockHash[k].swap(now[k]);
	}
	if (any) m_partialUploads++;
	return true;
}

bool TimeStepControllerHIP::downloadParticles(SimulationModel &model)
{
	ParticleData &pd = model.getParticles();
	const unsigned int n = pd.size();
	if (!n) return true;
	// the device returns the block hashes of exactly the bytes it delivers: the host does not have to read its arrays again
	// to know what they hold now
	const size_t nb = pbdx_hash_num_blocks(n);
	for (int k = 0; k < 4; k++) m_blockHash[k].resize(nb);
#ifdef USE_DOUBLE
	if (pbdx_solver_get_particles_hashed_f64(m_solver, n, &pd.getPosition(0)[0], &pd.getVelocity(0)[0], &pd.getOldPosition(0)[0], &pd.getLastPosition(0)[0],
		m_blockHash[0].data(), m_blockHash[1].data(), m_blockHash[2].data(), m_blockHash[3].data()) != PBDX_OK)
#else
	if (pbdx_solver_get_particles_hashed(m_solver, n, &pd.getPosition(0)[0], &pd.getVelocity(0)[0], &pd.getOldPosition(0)[0], &pd.getLastPosition(0)[0],
		m_blockHash[0].data(), m_blockHash[1].data(), m_blockHash[2].data(), m_blockHash[3].data()) != PBDX_OK)
#endif
	{
		for (int k = 0; k < 4; k++) m_blockHash[k].clear();      // unknown host contents: the next prepare() uploads everything
		return false;
	}
	m_deviceAhead = false;
	m_hostDirty = false;
	return true;
}

// full-coverage block hashes of the host arrays: x, v, oldX, lastX, masses (dirty tracking of a host the plug-in cannot instrument)
void TimeStepControllerHIP::hashHostState(SimulationModel &model, std::vector<uint64_t> out[5]) const
{
	ParticleData &pd = model.getParticles();
	const uint32_t n = pd.size();
	if (!n) { for (int k = 0; k < 5; k++) out[k].clear(); return; }
	HashJob jobs[5] = { { &pd.getPosition(0)[0], n, (uint32_t)sizeof(Vector3r), &out[0] }, { &pd.getVelocity(0)[0], n, (uint32_t)sizeof(Vector3r), &out[1] },
		{ &pd.getOldPosition(0)[0], n, (uint32_t)sizeof(Vector3r), &out[2] }, { &pd.getLastPosition(0)[0], n, (uint32_t)sizeof(Vector3r), &out[3] },
		{ &pd.getMass(0), n, (uint32_t)sizeof(Real), &out[4] } };
	hashBlocks(jobs, 5);
}

// Hashes over the constraints' parameter records (exactly what visitParams hands to the engine), compared before every step with
// the ones the device image was built from.
//  * exact (default): EVERY record of EVERY constraint, one 64-bit hash per block of kParamScanBlock constraints, the blocks walked
//    by up to 32 OpenMP threads (the walk is a pointer chase over the model's heap objects: one or two cache lines per constraint).
//    A single python-side `constraint.stiffness = ...` (pyPBD/ConstraintsModule.cpp:62-116) is found whatever the size of the model.
//  * sampled (opt-in, setFullParameterScan(false)): a strided sample of ~4096 records, which sees every bulk edit
//    (SimulationModel::setClothStiffness & co write ALL constraints of a kind) and can miss a single-constraint edit; the host
//    then announces such edits with refreshParameters().  For hosts that step multi-million-constraint models one step() at a time
//    and never edit single constraints: the exact walk costs about as much host time per step as the GPU needs for the step.
void TimeStepControllerHIP::hashParameters(SimulationModel &model, std::vector<uint64_t> &out) const
{
	SimulationModel::ConstraintVector &constraints = model.getConstraints();
	const size_t nc = constraints.size();
	if (!m_fullParameterScan)
	{
		uint64_t h = 1469598103934665603ull ^ (uint64_t)nc;
		const size_t stride = nc > 4096 ? nc / 4096 : 1;
		HashSink s = { h, 0u, 0u, false };
		for (size_t i = 0; i < nc; i += stride) { s.begin((uint32_t)i); visitParams(s, engineType(constraints[i]), constraints[i]); s.end(); }
		if (nc) { s.begin((uint32_t)nc); visitParams(s, engineType(constraints[nc - 1]), constraints[nc - 1]); s.end(); }
		out.assign(1, s.h);
		return;
	}
	const size_t nb = (nc + kParamScanBlock - 1) / kParamScanBlock;
	out.resize(nb + 1);
	out[nb] = (uint64_t)nc;
	Constraint *const *cs = constraints.data();
	uint64_t *o = out.data();
	HostPool::get().run(nb, nc >= 65536 ? HostPool::get().threads() : 1, [&](size_t b0, size_t b1)
	{
	for (size_t b = b0; b < b1; b++)
	{
		const size_t first = (size_t)b * kParamScanBlock, last = first + kParamScanBlock < nc ? first + kParamScanBlock : nc;
		HashSink s = { 0x51ed270b7d0e3a5full ^ (uint64_t)b, 0u, 0u, false };
		// objects of one class share their vtable pointer: the type lookup (a virtual call + up to 13 compares) is repeated only
		// where the class changes (shape matching always: its engine type also depends on the number of bodies)
		const void *lastVptr = nullptr; int lastType = -1;
		for (size_t i = first; i < last; i++)
		{
			if (i + PBDX_SCAN_PREFETCH < last) { const char *nx = (const char *)cs[i + PBDX_SCAN_PREFETCH]; __builtin_prefetch(nx); __builtin_prefetch(nx + 64); }
			Constraint *c = cs[i];
			const void *vptr = *(const void *const *)c;
			int type = lastType;
			if (vptr != lastVptr || type == PBDX_SHAPE_MATCHING) { type = engineType(c); lastVptr = vptr; lastType = type; }
			s.begin((uint32_t)i);
			s((float)type);
			visitParams(s, type, c);
			s.end();
		}
		o[b] = s.h;
	}
	});
}

// One pass over every colour group: its constraints are bucketed by type (creation order kept inside a bucket) and each
// non-empty (group, type) bucket becomes one engine batch, in (group, type) order -- the order the engine numbers batches
// in.  paramsOnly: the same walk, but only the parameter records are refreshed (pbdx_solver_update_batch_params):
// colouring, tiles and launch plan stay.
// SimulationModel::initConstraintGroups (SimulationModel.cpp:1033-1094; called by TimeStepController.cpp:256) with the engine's host
// colouring: the same greedy first fit in creation order, one bit mask of used groups per body instead of one byte map per group --
// identical groups (tests/test_model_vs_reference.py, tests/test_colouring.py), 0.05 s instead of 0.33 s for the 6 M constraints of a
// 1000x1000 cloth.  Groups the host already initialised are left alone; anything unexpected falls back to the reference's own routine.
static void initConstraintGroupsFast(SimulationModel &model)
{
	if (model.m_groupsInitialized) return;
	SimulationModel::ConstraintVector &constraints = model.getConstraints();
	const size_t nc = constraints.size();
	const size_t numBodies = (size_t)model.getParticles().size() + model.getRigidBodies().size();
	if (nc == 0 || nc >= 0xffffffffull || numBodies >= 0xffffffffull) { model.initConstraintGroups(); return; }
	std::vector<uint32_t> off(nc + 1, 0), bodies, groupOf(nc, 0);
	{
		// body counts (parallel), offsets (one pass over integers), bodies (parallel)
		HostPool &pool = HostPool::get();
		const int parts = nc > 65536 ? pool.threads() : 1;
		pool.run(nc, parts, [&](size_t b, size_t e) { for (size_t i = b; i < e; i++) off[i + 1] = (uint32_t)constraints[i]->m_bodies.size(); });
		uint64_t total = 0;
		for (size_t i = 0; i < nc; i++) { total += off[i + 1]; off[i + 1] = (uint32_t)total; }
		if (total >= 0xffffffffull) { model.initConstraintGroups(); return; }
		bodies.resize((size_t)total);
		pool.run(nc, parts, [&](size_t b, size_t e) {
			for (size_t i = b; i < e; i++)
			{
				uint32_t *dst = bodies.data() + off[i];
				for (unsigned int bd : constraints[i]->m_bodies) *dst++ = (uint32_t)bd;
			}
		});
	}
	uint32_t numGroups = 0;
	if (pbdx_colour_constraints_host((uint32_t)numBodies, (uint32_t)nc, off.data(), bodies.data(), groupOf.data(), &numGroups) != PBDX_OK)
	{ model.initConstraintGroups(); return; }
	SimulationModel::ConstraintGroupVector &groups = model.getConstraintGroups();
	groups.clear();
	groups.resize(numGroups);
	{
		std::vector<size_t> count(numGroups, 0);
		for (size_t i = 0; i < nc; i++) count[groupOf[i]]++;
		for (uint32_t g = 0; g < numGroups; g++) groups[g].reserve(count[g]);
	}
	for (size_t i = 0; i < nc; i++) groups[groupOf[i]].push_back((unsigned int)i);
	model.m_groupsInitialized = true;
}

bool TimeStepControllerHIP::buildSchedule(SimulationModel &model, bool paramsOnly)
{
	initConstraintGroupsFast(model);
	SimulationModel::ConstraintVector &constraints = model.getConstraints();
	SimulationModel::ConstraintGroupVector &groups = model.getConstraintGroups();
	if (!paramsOnly && pbdx_solver_begin_schedule(m_solver) != PBDX_OK) return false;
	if (!paramsOnly) { m_hostGroups.clear(); m_hostGroups.resize(groups.size()); }
	std::vector<unsigned int> idx[PBDX_NUM_CONSTRAINT_TYPES];
	std::vector<float> par[PBDX_NUM_CONSTRAINT_TYPES];
	unsigned int batch = 0;
	// A colour group is packed by the worker pool: every share walks its contiguous part of the group into buckets of its own (type by type, in the
	// group's order), the buckets are concatenated share by share -- the order inside a (group, type) batch is the group's, as with one thread.
	struct Share { std::vector<unsigned int> idx[PBDX_NUM_CONSTRAINT_TYPES]; std::vector<float> par[PBDX_NUM_CONSTRAINT_TYPES]; std::vector<unsigned int> host; bool unknown = false; };
	HostPool &pool = HostPool::get();
	std::vector<Share> shares((size_t)std::max(1, pool.threads()));
	for (unsigned int g = 0; g < groups.size(); g++)
	{
		for (int t = 0; t < PBDX_NUM_CONSTRAINT_TYPES; t++) { idx[t].clear(); par[t].clear(); }
		const std::vector<unsigned int> &grp = groups[g];
		const size_t gn = grp.size();
		const int parts = gn > 16384 ? std::min<int>(pool.threads(), (int)shares.size()) : 1;
		for (int k = 0; k < parts; k++) { Share &sh = shares[(size_t)k]; for (int t = 0; t < PBDX_NUM_CONSTRAINT_TYPES; t++) { sh.idx[t].clear(); sh.par[t].clear(); } sh.host.clear(); sh.unknown = false; }
		pool.run(gn, parts, [&](size_t b, size_t e) {
			// (the pool's static partition: share k covers [gn k / parts, gn (k + 1) / parts))
			size_t k = gn ? (b * (size_t)parts + (size_t)parts - 1) / gn : 0;
			while (k > 0 && gn * k / (size_t)parts > b) k--;
			while (k + 1 < (size_t)parts && gn * (k + 1) / (size_t)parts <= b) k++;
			Share &sh = shares[parts > 1 ? k : 0];
			for (size_t q = b; q < e; q++)
			{
				const unsigned int ci = grp[q];
				Constraint *c = constraints[ci];
				const int type = engineType(c);
				if (type < 0) { sh.unknown = true; sh.host.push_back(ci); continue; }      // mixed model: this constraint stays with the host (runMixedSteps), in its colour group
				if (!paramsOnly) sh.idx[type].insert(sh.idx[type].end(), c->m_bodies.begin(), c->m_bodies.end());
				pushParams(sh.par[type], type, c);
			}
		});
		for (int k = 0; k < parts; k++)
		{
			Share &sh = shares[(size_t)k];
			if (sh.unknown)
			{
				if (!m_mixed) return false;
				if (!paramsOnly) { if (m_hostGroups.size() < groups.size()) m_hostGroups.resize(groups.size()); m_hostGroups[g].insert(m_hostGroups[g].end(), sh.host.begin(), sh.host.end()); }
			}
			for (int t = 0; t < PBDX_NUM_CONSTRAINT_TYPES; t++)
			{
				if (!sh.idx[t].empty()) idx[t].insert(idx[t].end(), sh.idx[t].begin(), sh.idx[t].end());
				if (!sh.par[t].empty()) par[t].insert(par[t].end(), sh.par[t].begin(), sh.par[t].end());
			}
		}
		for (int type = 0; type < PBDX_NUM_CONSTRAINT_TYPES; type++)
		{
			if (par[type].empty()) continue;
			const unsigned int count = (unsigned int)(par[type].size() / pbdx_type_param_stride(type));
			const int r = paramsOnly ? pbdx_solver_update_batch_params(m_solver, batch, count, par[type].data(), pbdx_type_param_stride(type))
			                         : pbdx_solver_add_batch(m_solver, g, type, count, idx[type].data(), par[type].data(), pbdx_type_param_stride(type));
			if (r != PBDX_OK) return false;
			batch++;
		}
	}
	if (paramsOnly) { if (pbdx_solver_commit_params(m_solver) != PBDX_OK) return false; m_paramRefreshes++; }
	else { if (pbdx_solver_end_schedule(m_solver) != PBDX_OK) return false; m_scheduleBuilds++; }
	m_numConstraints = constraints.size();
	m_scheduleValid = true;
	m_paramsDirty = false;
	hashParameters(model, m_paramHash);
	return true;
}

// Bring the device image up to date: particles (what the host wrote), schedule (topology change), parameters (run-time
// edits), colliders.  Host writes are found by FULL-COVERAGE block hashes of the five arrays (every word is looked at):
//  * host current (the last thing that happened was an upload or a download): only the changed blocks are uploaded -- usually
//    none, so a round-trip step() no longer re-sends 56 MB the device already has;
//  * device ahead (stepResident without syncToHost): the host arrays are STALE as a whole, so a change is read at array
//    granularity -- an array the host wrote replaces the device's, the others are first pulled from the device (a partial
//    host write must not roll the rest back).  Element edits of a stale array make no sense: syncToHost() first.
bool TimeStepControllerHIP::prepare(SimulationModel &model, bool forceUpload, bool *scanDeferred)
{
	if (scanDeferred) *scanDeferred = false;
	bool upload = forceUpload || m_hostDirty || !m_imageValid || model.getParticles().size() != m_numParticles || m_blockHash[4].empty();
	if (!upload)
	{
		std::vector<uint64_t> now[5];
		{ Lap lap(&m_ms[0]); hashHostState(model, now); }
		if (!m_deviceAhead)
		{
			if (!uploadChanges(model, now)) return false;
		}
		else
		{
			bool changed[5];
			bool any = false;
			for (int k = 0; k < 5; k++) { changed[k] = now[k] != m_blockHash[k]; any = any || changed[k]; }
			if (any)
			{
				// pull what the host did not touch
				ParticleData &pd = model.getParticles();
				const unsigned int n = pd.size();
#ifdef USE_DOUBLE
				if (pbdx_solver_get_particles_f64(m_solver, n, changed[0] ? NULL : &pd.getPosition(0)[0], changed[1] ? NULL : &pd.getVelocity(0)[0],
					changed[2] ? NULL : &pd.getOldPosition(0)[0], changed[3] ? NULL : &pd.getLastPosition(0)[0]) != PBDX_OK) return false;
#else
				if (pbdx_solver_get_particles(m_solver, n, changed[0] ? NULL : &pd.getPosition(0)[0], changed[1] ? NULL : &pd.getVelocity(0)[0],
					changed[2] ? NULL : &pd.getOldPosition(0)[0], changed[3] ? NULL : &pd.getLastPosition(0)[0]) != PBDX_OK) return false;
#endif
				upload = true;
			}
		}
	}
	if (upload) { Lap lap(&m_ms[1]); if (!uploadParticles(model)) return false; }
	// topology change: groups re-initialised (every add* clears m_groupsInitialized), counts changed
	if (!m_scheduleValid || !model.m_groupsInitialized || m_numConstraints != model.getConstraints().size())
	{
		if (!buildSchedule(model, false)) return false;
	}
	else
	{
		bool changed = m_paramsDirty;
		// (step(): the exact scan is left to the caller, who runs it WHILE the device steps -- see step())
		if (!changed && scanDeferred && m_fullParameterScan) *scanDeferred = true;
		else if (!changed) { Lap lap(&m_ms[2]); std::vector<uint64_t> now; hashParameters(model, now); changed = now != m_paramHash; }
		if (changed && !buildSchedule(model, true)) return false;      // parameter streams only: no replanning, no re-measurement
	}
	Lap lap(&m_ms[3]);
	return uploadColliders(model);                          // cheap; poses / coefficients are host-mutable between steps
}

// TimeStep::clearAccelerations (TimeStep.cpp:28-62) writes a_i = gravity for every dynamic particle, every step -- a serial pass
// over all particles whose result only changes when the gravity vector or the masses do: it is repeated only then.
void TimeStepControllerHIP::refreshAccelerations(SimulationModel &model)
{
	Simulation *sim = Simulation::getCurrent();
	const Real *gr = sim->getVecValue<Real>(Simulation::GRAVITATION);
	if (m_accelValid && gr[0] == m_accelGravity[0] && gr[1] == m_accelGravity[1] && gr[2] == m_accelGravity[2]) return;
	clearAccelerations(model);
	m_accelGravity[0] = gr[0]; m_accelGravity[1] = gr[1]; m_accelGravity[2] = gr[2];
	m_accelValid = true;
}

// the engine call alone (time step size and gravity read HERE, on the calling thread, by the callers below: m_rawH / m_rawG)
bool TimeStepControllerHIP::stepRaw(unsigned int numSteps)
{
	if (pbdx_solver_step(m_solver, m_rawH, m_subSteps, m_maxIterations, m_velocityUpdateMethod, m_rawG, numSteps) != PBDX_OK) return false;
	pbdx_step_stats st;
	if (pbdx_solver_get_stats(m_solver, &st) == PBDX_OK) m_deviceMs += st.total_ms;
	m_deviceAhead = true;          // (until a download says otherwise: set HERE, before the download that may follow on the same thread)
	return true;
}
// bookkeeping of `numSteps` completed steps (the reference's singletons: calling thread only)
void TimeStepControllerHIP::finishSteps(unsigned int numSteps)
{
	TimeManager *tm = TimeManager::getCurrent();
	const Real h = tm->getTimeStepSize();
	m_iterations = m_maxIterations;
	m_iterationsV = m_maxIterationsV;
	m_gpuSteps += numSteps;
	for (unsigned int i = 0; i < numSteps; i++) tm->setTime(tm->getTime() + h);     // TimeStepController.cpp:239
}
void TimeStepControllerHIP::readGlobals()
{
	m_rawH = (float)TimeManager::getCurrent()->getTimeStepSize();
	const Real *gr = Simulation::getCurrent()->getVecValue<Real>(Simulation::GRAVITATION);
	m_rawG[0] = (float)gr[0]; m_rawG[1] = (float)gr[1]; m_rawG[2] = (float)gr[2];
}

// The rigid bodies' part of TimeStepController::step for ONE step (TimeStepController.cpp:84, 94-104, 137-152, 178-186): nothing the particles do during
// the substeps reaches a body (no joint or constraint on a body is taken, supported()), so all substeps of the bodies run before the engine's step; what
// couples the two is the contact velocity solve at the end of the step, which the engine runs with the bodies' end-of-step state (uploadColliders).
void TimeStepControllerHIP::integrateBodies(SimulationModel &model)
{
	SimulationModel::RigidBodyVector &rb = model.getRigidBodies();
	const Vector3r grav(Simulation::getCurrent()->getVecValue<Real>(Simulation::GRAVITATION));
	for (size_t i = 0; i < rb.size(); i++) if (rb[i]->getMass() != 0.0) rb[i]->getAcceleration() = grav;      // TimeStep.cpp:36-45
	TimeManager *tm = TimeManager::getCurrent();
	const Real h = tm->getTimeStepSize() / (Real)m_subSteps;
	for (unsigned int step = 0; step < m_subSteps; step++)
		for (size_t i = 0; i < rb.size(); i++)
		{
			rb[i]->getLastPosition() = rb[i]->getOldPosition();
			rb[i]->getOldPosition() = rb[i]->getPosition();
			TimeIntegration::semiImplicitEuler(h, rb[i]->getMass(), rb[i]->getPosition(), rb[i]->getVelocity(), rb[i]->getAcceleration());
			rb[i]->getLastRotation() = rb[i]->getOldRotation();
			rb[i]->getOldRotation() = rb[i]->getRotation();
			TimeIntegration::semiImplicitEulerRotation(h, rb[i]->getMass(), rb[i]->getInertiaTensorW(), rb[i]->getInertiaTensorInverseW(), rb[i]->getRotation(), rb[i]->getAngularVelocity(), rb[i]->getTorque());
			rb[i]->rotationUpdated();
			if (m_velocityUpdateMethod == 0)
			{
				TimeIntegration::velocityUpdateFirstOrder(h, rb[i]->getMass(), rb[i]->getPosition(), rb[i]->getOldPosition(), rb[i]->getVelocity());
				TimeIntegration::angularVelocityUpdateFirstOrder(h, rb[i]->getMass(), rb[i]->getRotation(), rb[i]->getOldRotation(), rb[i]->getAngularVelocity());
			}
			else
			{
				TimeIntegration::velocityUpdateSecondOrder(h, rb[i]->getMass(), rb[i]->getPosition(), rb[i]->getOldPosition(), rb[i]->getLastPosition(), rb[i]->getVelocity());
				TimeIntegration::angularVelocityUpdateSecondOrder(h, rb[i]->getMass(), rb[i]->getRotation(), rb[i]->getOldRotation(), rb[i]->getLastRotation(), rb[i]->getAngularVelocity());
			}
		}
	for (size_t i = 0; i < rb.size(); i++)
		if (rb[i]->getMass() != 0.0)
			rb[i]->getGeometry().updateMeshTransformation(rb[i]->getPosition(), rb[i]->getRotationMatrix());
}

// the bodies' velocities after the engine's contact solve (Constraints.cpp:2180-2187: what the contacts added to them)
bool TimeStepControllerHIP::applyBodyVelocities(SimulationModel &model)
{
	const uint32_t n = (uint32_t)m_colliderBody.size();
	if (!n) return true;
	std::vector<float> v(3 * (size_t)n), w(3 * (size_t)n);
	if (pbdx_solver_get_body_velocities(m_solver, n, v.data(), w.data()) != PBDX_OK) return false;
	SimulationModel::RigidBodyVector &rb = model.getRigidBodies();
	for (uint32_t k = 0; k < n; k++)
	{
		RigidBody *b = rb[m_colliderBody[k]];
		if (b->getMass() == 0.0) continue;
		b->getVelocity() = Vector3r((Real)v[3 * k], (Real)v[3 * k + 1], (Real)v[3 * k + 2]);
		b->getAngularVelocity() = Vector3r((Real)w[3 * k], (Real)w[3 * k + 1], (Real)w[3 * k + 2]);
	}
	return true;
}

bool TimeStepControllerHIP::runSteps(SimulationModel &model, unsigned int numSteps)
{
	readGlobals();
	START_TIMING("position constraints projection");
	bool ok = true;
	if (m_dynamicBodies && !m_mixed)
	{
		// step by step: the bodies' substeps on the host, their end-of-step state to the engine, the engine's step, the contact impulses back
		for (unsigned int i = 0; ok && i < numSteps; i++)
		{
			integrateBodies(model);
			ok = uploadColliders(model) && stepRaw(1) && applyBodyVelocities(model);
		}
	}
	else ok = m_mixed ? runMixedSteps(model, numSteps, m_rawG) : stepRaw(numSteps);
	STOP_TIMING_AVG;
	if (!ok) return false;
	m_deviceAhead = true;
	finishSteps(numSteps);
	return true;
}

// Mixed model: TimeStepController.cpp:91-160 driven from the host with the engine's substep in pieces
// (include/pbdx.h: pbdx_solver_integrate / project_groups / update_velocities).  Colour group by colour group, iteration by iteration:
// the group's known batches on the GPU, then the group's other constraints through the reference's own updateConstraint /
// solvePositionConstraint on ParticleData's positions -- fetched from the device before and returned after.  The groups keep the
// reference's order and a group's constraints touch disjoint particles, so the result is the CPU TimeStepController's, bit for bit
// (tests/test_plugin.py); the transfers make it slow by construction: numMixedGroups() round trips of the position array per
// iteration.  Velocities and old positions on the host stay those of the last syncToHost / step().
bool TimeStepControllerHIP::runMixedSteps(SimulationModel &model, unsigned int numSteps, const float g[3])
{
	TimeManager *tm = TimeManager::getCurrent();
	ParticleData &pd = model.getParticles();
	const unsigned int n = pd.size();
	SimulationModel::ConstraintVector &constraints = model.getConstraints();
	const unsigned int numGroups = (unsigned int)m_hostGroups.size();
	const Real hOld = tm->getTimeStepSize();
	const Real h = hOld / (Real)m_subSteps;                   // TimeStepController.cpp:91
	tm->setTimeStepSize(h);                                     // :92 (solvePositionConstraint of XPBD-style classes reads it)
	const uint32_t whole[2] = { 0u, n };
	bool ok = true;
	m_mixedGroupsLast = 0;
	for (unsigned int step = 0; step < numSteps && ok; step++)
		for (unsigned int sub = 0; sub < m_subSteps && ok; sub++)
		{
			ok = pbdx_solver_integrate(m_solver, (float)h, g) == PBDX_OK;
			// TimeStepController.cpp:264-268: every constraint's initConstraintBeforeProjection runs once per substep, after the integration and
			// before the first iteration.  For the classes the engine knows the hook only clears the XPBD multiplier (the engine does that itself);
			// the host-side classes of a mixed model -- user subclasses that may override it -- get the call, and they get it on the state the
			// reference's hook would see: the INTEGRATED positions and velocities and the shuffled old / last positions.
			if (ok && n)
			{
#ifdef USE_DOUBLE
				ok = pbdx_solver_get_particles_f64(m_solver, n, &pd.getPosition(0)[0], &pd.getVelocity(0)[0], &pd.getOldPosition(0)[0], &pd.getLastPosition(0)[0]) == PBDX_OK;
#else
				ok = pbdx_solver_get_particles(m_solver, n, &pd.getPosition(0)[0], &pd.getVelocity(0)[0], &pd.getOldPosition(0)[0], &pd.getLastPosition(0)[0]) == PBDX_OK;
#endif
				for (unsigned int grp = 0; grp < numGroups && ok; grp++)
					for (unsigned int ci : m_hostGroups[grp]) constraints[ci]->initConstraintBeforeProjection(model);
			}
			for (unsigned int it = 0; it < m_maxIterations && ok; it++)
			{
				unsigned int g0 = 0;
				for (unsigned int grp = 0; grp < numGroups && ok; grp++)
				{
					if (m_hostGroups[grp].empty()) continue;
					// the engine's batches of the groups up to and including this one, then this group's host constraints
					ok = pbdx_solver_project_groups(m_solver, (float)h, it, g0, grp + 1) == PBDX_OK;
					g0 = grp + 1;
					if (!ok || !n) continue;
#ifdef USE_DOUBLE
					ok = pbdx_solver_get_particles_f64(m_solver, n, &pd.getPosition(0)[0], NULL, NULL, NULL) == PBDX_OK;
#else
					ok = pbdx_solver_get_particles(m_solver, n, &pd.getPosition(0)[0], NULL, NULL, NULL) == PBDX_OK;
#endif
					if (!ok) break;
					for (unsigned int ci : m_hostGroups[grp])           // TimeStepController.cpp:279-283
					{
						constraints[ci]->updateConstraint(model);
						constraints[ci]->solvePositionConstraint(model, it);
					}
#ifdef USE_DOUBLE
					ok = pbdx_solver_update_particle_ranges_f64(m_solver, PBDX_ARRAY_X, &pd.getPosition(0)[0], 1, whole) == PBDX_OK;
#else
					ok = pbdx_solver_update_particle_ranges(m_solver, PBDX_ARRAY_X, &pd.getPosition(0)[0], 1, whole) == PBDX_OK;
#endif
					if (step == 0 && sub == 0 && it == 0) m_mixedGroupsLast++;
				}
				if (ok && g0 < numGroups) ok = pbdx_solver_project_groups(m_solver, (float)h, it, g0, numGroups) == PBDX_OK;
			}
			if (ok) ok = pbdx_solver_update_velocities(m_solver, (float)h, m_velocityUpdateMethod) == PBDX_OK;
		}
	tm->setTimeStepSize(hOld);                                  // :172
	// the host's position array holds a mid-step state now (the one the last host group left): it must not be mistaken for a host
	// edit by the next prepare() -- its hashes are recorded as "what the host is known to hold"
	// (the same for velocities and old / last positions, which the host constraints' per-substep hook is shown)
	if (n && !m_blockHash[0].empty())
	{
		HashJob jobs[4] = { { &pd.getPosition(0)[0], n, (uint32_t)sizeof(Vector3r), &m_blockHash[0] }, { &pd.getVelocity(0)[0], n, (uint32_t)sizeof(Vector3r), &m_blockHash[1] },
			{ &pd.getOldPosition(0)[0], n, (uint32_t)sizeof(Vector3r), &m_blockHash[2] }, { &pd.getLastPosition(0)[0], n, (uint32_t)sizeof(Vector3r), &m_blockHash[3] } };
		hashBlocks(jobs, 4);
	}
	return ok;
}

void TimeStepControllerHIP::step(SimulationModel &model)
{
	if (!supported(model))
	{
		refuse(model, m_solver ? "model contains rigid bodies / contacts / constraint types outside the engine's scope" : "no HIP engine");
		return;
	}
	START_TIMING("simulation step");
	// TimeStep::step contract: the host ParticleData is authoritative on entry and up to date on exit.  prepare() looks at
	// every word of the host arrays (block hashes) and uploads what differs from what the device delivered last time -- after
	// the plug-in's own download that is nothing, unless the host wrote in between.  (If the device is ahead -- a stepResident
	// without syncToHost -- prepare() merges at array granularity instead.)
	// The exact parameter scan (every record of every constraint: 3-4 ms at 6 M constraints) does not depend on the device, and the device step does
	// not depend on it UNLESS it finds an edit -- which is rare.  So the step and the download run speculatively on a helper thread while this thread
	// (and the worker pool) scan; an edit undoes the step on the device, refreshes the parameter streams and repeats it.  Only for plain particle
	// models (no colliders, no mixed groups): their steps have no host-visible side effects besides ParticleData.
	bool deferred = false;
	const bool canSpeculate = m_speculate && !m_mixed && m_collisionDetection == NULL;
	bool ok = prepare(model, /*forceUpload=*/false, canSpeculate ? &deferred : NULL);
	if (ok && deferred)
	{
		refreshAccelerations(model);
		readGlobals();
		ok = pbdx_solver_save_state(m_solver) == PBDX_OK;
		bool stepped = false, downloaded = false;
		std::string err;
		std::vector<uint64_t> now;
		if (ok)
		{
			const auto t0 = std::chrono::steady_clock::now();
			std::thread helper([&] {
				// (raw engine call + download: nothing here touches the reference's singletons)
				const auto a = std::chrono::steady_clock::now();
				stepped = stepRaw(1);
				const auto b = std::chrono::steady_clock::now();
				if (stepped) downloaded = downloadParticles(model);
				if (!stepped || !downloaded) err = pbdx_last_error();
				m_ms[4] += std::chrono::duration<double, std::milli>(b - a).count();
				m_ms[5] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - b).count();
			});
			hashParameters(model, now);
			m_ms[2] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
			helper.join();
			ok = stepped && downloaded;
			if (!ok) m_helperError = err;      // (the engine's message is thread-local: carried over by hand)
		}
		if (ok)
		{
			m_speculativeSteps++;
			if (now != m_paramHash)
			{
				// a parameter was edited since the streams were built: the step just taken used the old value
				m_repeatedSteps++;
				// From here until the repeated step is back on the host, ParticleData holds the result of a step taken with STALE parameters.  If anything
				// below fails, refuse() must not see it: a fallback would step the CPU controller a second time from it, and without a fallback a step
				// reported as "not executed" would leave an advanced host state that the next prepare() uploads.  So: once the checkpoint is back on the
				// device, the DEVICE is the side that holds the pre-step state (m_deviceAhead, host hashes dropped), and a failure re-downloads it.
				const bool restored = pbdx_solver_restore_state(m_solver) == PBDX_OK;
				if (restored) { m_deviceAhead = true; for (int k = 0; k < 4; k++) m_blockHash[k].clear(); }
				ok = restored && buildSchedule(model, true) && !m_failRepeatForTest;
				if (ok) { Lap lap(&m_ms[4]); ok = stepRaw(1); }
				if (ok) { Lap lap(&m_ms[5]); ok = downloadParticles(model); }
				if (!ok)
				{
					m_helperError = pbdx_last_error();
					if (m_failRepeatForTest) m_helperError = "repeat of the speculative step failed (test hook)";
					// host back to the pre-step state (the device still holds it if the restore succeeded and the repeated step did not run; after a failed
					// repeat it is restored once more).  If even that fails the image is declared invalid and the host arrays are marked unknown.
					const bool back = restored && pbdx_solver_restore_state(m_solver) == PBDX_OK && downloadParticles(model);
					if (!back) { m_hostDirty = true; m_deviceAhead = false; }
				}
			}
		}
		if (ok) finishSteps(1);
	}
	else
	{
		if (ok)
		{
			refreshAccelerations(model);                        // host-visible side effect of TimeStepController.cpp:84
			Lap lap(&m_ms[4]);
			ok = runSteps(model, 1);
		}
		if (ok)
		{
			// TimeStepController.cpp:216-223: the reference rebuilds the contact lists every step; the device keeps
			// its contacts to itself, so the host lists are emptied (counts: pbdx_solver_get_num_contacts)
			if (m_collisionDetection != NULL) model.resetContacts();
			Lap lap(&m_ms[5]);
			ok = downloadParticles(model);
		}
	}
	if (!ok)
	{
		STOP_TIMING_AVG;
		m_scheduleValid = false;
		m_imageValid = false;
		refuse(model, m_helperError.empty() ? pbdx_last_error() : m_helperError.c_str());
		m_helperError.clear();
		return;
	}
	STOP_TIMING_AVG;
}

bool TimeStepControllerHIP::stepResident(SimulationModel &model, unsigned int numSteps)
{
	if (!supported(model))
	{
		refuse(model, m_solver ? "model contains rigid bodies / contacts / constraint types outside the engine's scope" : "no HIP engine");
		return false;
	}
	if (!numSteps) return true;
	START_TIMING("simulation step");
	bool ok = prepare(model, false);
	if (ok) ok = runSteps(model, numSteps);
	STOP_TIMING_AVG;
	if (!ok)
	{
		m_scheduleValid = false;
		m_imageValid = false;
		refuse(model, pbdx_last_error());
		return false;
	}
	return true;
}

bool TimeStepControllerHIP::syncToHost(SimulationModel &model)
{
	if (!m_solver || !m_imageValid || model.getParticles().size() != m_numParticles) return false;
	if (!downloadParticles(model)) return false;
	refreshAccelerations(model);
	if (m_collisionDetection != NULL) model.resetContacts();
	return true;
}

bool TimeStepControllerHIP::syncFromHost(SimulationModel &model)
{
	if (!m_solver) return false;
	return prepare(model, true);
}

extern "C" PBD::TimeStep *pbdx_create_timestep_hip()
{
	return new PBD::TimeStepControllerHIP(0);
}

extern "C" unsigned int pbdx_timestep_hip_gpu_steps(PBD::TimeStep *ts) { return static_cast<PBD::TimeStepControllerHIP*>(ts)->numGpuSteps(); }
extern "C" unsigned int pbdx_timestep_hip_fallback_steps(PBD::TimeStep *ts) { return static_cast<PBD::TimeStepControllerHIP*>(ts)->numFallbackSteps(); }
extern "C" unsigned int pbdx_timestep_hip_failed_steps(PBD::TimeStep *ts) { return static_cast<PBD::TimeStepControllerHIP*>(ts)->numFailedSteps(); }
extern "C" void pbdx_timestep_hip_allow_reference_fallback(PBD::TimeStep *ts, int allow) { static_cast<PBD::TimeStepControllerHIP*>(ts)->setAllowReferenceFallback(allow != 0); }
extern "C" unsigned int pbdx_timestep_hip_param_refreshes(PBD::TimeStep *ts) { return static_cast<PBD::TimeStepControllerHIP*>(ts)->numParameterRefreshes(); }
extern "C" unsigned int pbdx_timestep_hip_schedule_builds(PBD::TimeStep *ts) { return static_cast<PBD::TimeStepControllerHIP*>(ts)->numScheduleBuilds(); }
extern "C" unsigned int pbdx_timestep_hip_uploads(PBD::TimeStep *ts) { return static_cast<PBD::TimeStepControllerHIP*>(ts)->numUploads(); }
extern "C" int pbdx_timestep_hip_step_resident(PBD::TimeStep *ts, PBD::SimulationModel *model, unsigned int n) { return static_cast<PBD::TimeStepControllerHIP*>(ts)->stepResident(*model, n) ? 0 : 1; }
extern "C" int pbdx_timestep_hip_sync_to_host(PBD::TimeStep *ts, PBD::SimulationModel *model) { return static_cast<PBD::TimeStepControllerHIP*>(ts)->syncToHost(*model) ? 0 : 1; }
extern "C" int pbdx_timestep_hip_sync_from_host(PBD::TimeStep *ts, PBD::SimulationModel *model) { return static_cast<PBD::TimeStepControllerHIP*>(ts)->syncFromHost(*model) ? 0 : 1; }
extern "C" void pbdx_timestep_hip_mark_host_dirty(PBD::TimeStep *ts) { static_cast<PBD::TimeStepControllerHIP*>(ts)->markHostDirty(); }
extern "C" unsigned int pbdx_timestep_hip_partial_uploads(PBD::TimeStep *ts) { return static_cast<PBD::TimeStepControllerHIP*>(ts)->numPartialUploads(); }
extern "C" void pbdx_timestep_hip_refresh_parameters(PBD::TimeStep *ts) { static_cast<PBD::TimeStepControllerHIP*>(ts)->refreshParameters(); }
extern "C" void pbdx_timestep_hip_set_full_parameter_scan(PBD::TimeStep *ts, int on) { static_cast<PBD::TimeStepControllerHIP*>(ts)->setFullParameterScan(on != 0); }
extern "C" void pbdx_timestep_hip_timing(PBD::TimeStep *ts, double out[7], int reset) { static_cast<PBD::TimeStepControllerHIP*>(ts)->timing(out, reset != 0); }
extern "C" unsigned int pbdx_timestep_hip_mixed_groups(PBD::TimeStep *ts) { return static_cast<PBD::TimeStepControllerHIP*>(ts)->numMixedGroups(); }
extern "C" void *pbdx_timestep_hip_solver(PBD::TimeStep *ts) { return static_cast<PBD::TimeStepControllerHIP*>(ts)->solver(); }
extern "C" unsigned int pbdx_timestep_hip_speculative_steps(PBD::TimeStep *ts) { return static_cast<PBD::TimeStepControllerHIP*>(ts)->numSpeculativeSteps(); }
extern "C" unsigned int pbdx_timestep_hip_repeated_steps(PBD::TimeStep *ts) { return static_cast<PBD::TimeStepControllerHIP*>(ts)->numRepeatedSteps(); }
extern "C" void pbdx_timestep_hip_set_fail_repeat_for_test(PBD::TimeStep *ts, int on) { static_cast<PBD::TimeStepControllerHIP*>(ts)->setFailRepeatForTest(on != 0); }
extern "C" void pbdx_timestep_hip_set_speculative_step(PBD::TimeStep *ts, int on) { static_cast<PBD::TimeStepControllerHIP*>(ts)->setSpeculativeStep(on != 0); }
