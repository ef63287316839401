/* oracle/pbd_oracle_impl.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the hot path of
 * InteractiveComputerGraphics/PositionBasedDynamics for particle scenes:
 * scene builders, greedy colouring, the 13 particle constraint projections and
 * TimeStepController::step.  Included twice by pbd_oracle.c with
 *   PO_REAL = float  / PO(name) = po32_name   (the reference's float build)
 *   PO_REAL = double / PO(name) = po64_name   (the reference's default double build)
 * Every function cites the reference file:line it follows; floating-point
 * expressions keep the association order Eigen 3.4 evaluates them in (3-term
 * reductions are c0 + (c1 + c2), see Eigen/src/Core/Redux.h) and the double
 * sub-expressions the reference has in a float build (1.0/h, sqrt(2.0*U) ...).
 *
 * Pinned against the real reference: tests/test_oracle_port.py checks this port
 * bit-for-bit (float) against oracle/_ref (the unmodified reference compiled
 * with -ffp-contract=off) and against the golden vectors under tests/golden/.
 * Nothing in the product may include, link or call this file.
 */

#ifndef PO_REAL
#error "include from pbd_oracle.c"
#endif

#define R PO_REAL
typedef struct { R x, y, z; } PO(v3);
#define V3 PO(v3)

static V3 PO(mk)(R x, R y, R z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
static V3 PO(add)(V3 a, V3 b) { return PO(mk)(a.x + b.x, a.y + b.y, a.z + b.z); }
static V3 PO(sub)(V3 a, V3 b) { return PO(mk)(a.x - b.x, a.y - b.y, a.z - b.z); }
static V3 PO(neg)(V3 a) { return PO(mk)(-a.x, -a.y, -a.z); }
static V3 PO(scl)(R s, V3 a) { return PO(mk)(s * a.x, s * a.y, s * a.z); }
static V3 PO(scr)(V3 a, R s) { return PO(mk)(a.x * s, a.y * s, a.z * s); }
static V3 PO(dvs)(V3 a, R s) { return PO(mk)(a.x / s, a.y / s, a.z / s); }
static R PO(dot)(V3 a, V3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
static R PO(sqn)(V3 a) { return a.x * a.x + (a.y * a.y + a.z * a.z); }
static R PO(nrm)(V3 a) { return PO_SQRT(PO(sqn)(a)); }
static V3 PO(crs)(V3 a, V3 b) { return PO(mk)(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static V3 PO(nmz)(V3 a) { R z = PO(sqn)(a); if (z > 0) { R s = PO_SQRT(z); return PO(dvs)(a, s); } return a; }

#define MK PO(mk)
#define ADD PO(add)
#define SUB PO(sub)
#define NEG PO(neg)
#define SCL PO(scl)
#define SCR PO(scr)
#define DVS PO(dvs)
#define DOT PO(dot)
#define SQN PO(sqn)
#define NRM PO(nrm)
#define CRS PO(crs)
#define NMZ PO(nmz)
#define EPS ((R)1e-6)

typedef struct { R m[3][3]; } PO(m3);
#define M3 PO(m3)

static M3 PO(mmul)(const M3 *A, const M3 *B)
{
	M3 r; int i, j;
	for (i = 0; i < 3; i++) for (j = 0; j < 3; j++)
		r.m[i][j] = A->m[i][0] * B->m[0][j] + (A->m[i][1] * B->m[1][j] + A->m[i][2] * B->m[2][j]);
	return r;
}
static M3 PO(mtr)(const M3 *A) { M3 r; int i, j; for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) r.m[i][j] = A->m[j][i]; return r; }
static V3 PO(mvec)(const M3 *A, V3 v)
{
	return MK(A->m[0][0] * v.x + (A->m[0][1] * v.y + A->m[0][2] * v.z),
	          A->m[1][0] * v.x + (A->m[1][1] * v.y + A->m[1][2] * v.z),
	          A->m[2][0] * v.x + (A->m[2][1] * v.y + A->m[2][2] * v.z));
}
/* Eigen determinant_impl<3> (LU/Determinant.h) */
static R PO(d3h)(const M3 *A, int a, int b, int c) { return A->m[0][a] * (A->m[1][b] * A->m[2][c] - A->m[1][c] * A->m[2][b]); }
static R PO(mdet)(const M3 *A) { return PO(d3h)(A, 0, 1, 2) - PO(d3h)(A, 1, 0, 2) + PO(d3h)(A, 2, 0, 1); }
/* Eigen compute_inverse<3> (LU/InverseImpl.h:125-175) */
static R PO(cof)(const M3 *A, int i, int j)
{
	int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
	return A->m[i1][j1] * A->m[i2][j2] - A->m[i1][j2] * A->m[i2][j1];
}
static M3 PO(minv)(const M3 *A)
{
	M3 r;
	R c0 = PO(cof)(A, 0, 0), c1 = PO(cof)(A, 1, 0), c2 = PO(cof)(A, 2, 0);
	R d = c0 * A->m[0][0] + (c1 * A->m[1][0] + c2 * A->m[2][0]);
	R id = (R)1 / d;
	r.m[1][2] = PO(cof)(A, 2, 1) * id; r.m[2][1] = PO(cof)(A, 1, 2) * id; r.m[2][2] = PO(cof)(A, 2, 2) * id;
	r.m[1][0] = PO(cof)(A, 0, 1) * id; r.m[1][1] = PO(cof)(A, 1, 1) * id; r.m[2][0] = PO(cof)(A, 0, 2) * id;
	r.m[0][0] = c0 * id; r.m[0][1] = c1 * id; r.m[0][2] = c2 * id;
	return r;
}
static V3 PO(col)(const M3 *A, int c) { return MK(A->m[0][c], A->m[1][c], A->m[2][c]); }
static void PO(setcol)(M3 *A, int c, V3 v) { A->m[0][c] = v.x; A->m[1][c] = v.y; A->m[2][c] = v.z; }
static V3 PO(row)(const M3 *A, int r) { return MK(A->m[r][0], A->m[r][1], A->m[r][2]); }
static void PO(setrow)(M3 *A, int r, V3 v) { A->m[r][0] = v.x; A->m[r][1] = v.y; A->m[r][2] = v.z; }

/* ------------------------------------------------------------------------- */
/* state */
typedef struct {
	int type;
	unsigned b[4];
	R p[24];          /* layout of include/pbdx.h */
	R lambda;
} PO(con);

typedef struct { unsigned off, nv, nf; unsigned *faces; unsigned ne; unsigned *edges; /* 4 per edge */ } PO(trimesh);
typedef struct { unsigned off, nv, nt; unsigned *tets; unsigned ne; unsigned *edges; /* 2 per edge */ unsigned *vtets; } PO(tetmesh);

typedef struct PO(sim_s) {
	unsigned n, cap;
	R *mass, *w, *x0, *x, *v, *a, *ox, *lx;
	unsigned nc, ccap; PO(con) *c;
	unsigned ntri; PO(trimesh) tri[64];
	unsigned ntet; PO(tetmesh) tet[64];
	unsigned ng; unsigned **grp; unsigned *gsz; int groups_ok;
	unsigned sub_steps, max_iter; int vel_method;
	R h, time; R g[3];
} PO(sim);

static void PO(free_groups)(PO(sim) *s)
{
	unsigned i;
	for (i = 0; i < s->ng; i++) free(s->grp[i]);
	free(s->grp); free(s->gsz); s->grp = NULL; s->gsz = NULL; s->ng = 0; s->groups_ok = 0;
}

PO(sim) *PO(create)(void)
{
	PO(sim) *s = (PO(sim) *)calloc(1, sizeof(PO(sim)));
	/* TimeStepController.cpp:23-32, Simulation.cpp:16, TimeManager.cpp:10 */
	s->sub_steps = 5; s->max_iter = 1; s->vel_method = 0;
	s->h = (R)0.005; s->time = 0; s->g[0] = 0; s->g[1] = (R)-9.81; s->g[2] = 0;
	return s;
}

void PO(destroy)(PO(sim) *s)
{
	unsigned i;
	if (!s) return;
	free(s->mass); free(s->w); free(s->x0); free(s->x); free(s->v); free(s->a); free(s->ox); free(s->lx); free(s->c);
	for (i = 0; i < s->ntri; i++) { free(s->tri[i].faces); free(s->tri[i].edges); }
	for (i = 0; i < s->ntet; i++) { free(s->tet[i].tets); free(s->tet[i].edges); free(s->tet[i].vtets); }
	PO(free_groups)(s);
	free(s);
}

static V3 PO(ld)(const R *a, unsigned i) { return MK(a[3 * i], a[3 * i + 1], a[3 * i + 2]); }
static void PO(st)(R *a, unsigned i, V3 v) { a[3 * i] = v.x; a[3 * i + 1] = v.y; a[3 * i + 2] = v.z; }

/* ParticleData::addVertex  ParticleData.h:128-138 */
unsigned PO(add_vertex)(PO(sim) *s, const double *p)
{
	unsigned i = s->n;
	if (s->n == s->cap)
	{
		unsigned nc = s->cap ? s->cap * 2 : 1024;
		s->mass = (R *)realloc(s->mass, nc * sizeof(R)); s->w = (R *)realloc(s->w, nc * sizeof(R));
		s->x0 = (R *)realloc(s->x0, 3 * nc * sizeof(R)); s->x = (R *)realloc(s->x, 3 * nc * sizeof(R));
		s->v = (R *)realloc(s->v, 3 * nc * sizeof(R)); s->a = (R *)realloc(s->a, 3 * nc * sizeof(R));
		s->ox = (R *)realloc(s->ox, 3 * nc * sizeof(R)); s->lx = (R *)realloc(s->lx, 3 * nc * sizeof(R));
		s->cap = nc;
	}
	{
		V3 q = MK((R)p[0], (R)p[1], (R)p[2]);
		PO(st)(s->x0, i, q); PO(st)(s->x, i, q); PO(st)(s->ox, i, q); PO(st)(s->lx, i, q);
		PO(st)(s->v, i, MK(0, 0, 0)); PO(st)(s->a, i, MK(0, 0, 0));
	}
	s->mass[i] = 1; s->w[i] = 1;
	s->n++;
	return i;
}

/* ParticleData::setMass  ParticleData.h:239-246 */
void PO(set_mass)(PO(sim) *s, unsigned i, double m)
{
	s->mass[i] = (R)m;
	s->w[i] = ((R)m != 0) ? (R)1 / (R)m : 0;
}

/* IndexedFaceMesh::buildNeighbors  Utils/IndexedFaceMesh.cpp:118-226 (per-vertex edge lists) */
static void PO(build_tri_edges)(PO(trimesh) *tm)
{
	unsigned f, j, k;
	unsigned **pe = (unsigned **)calloc(tm->nv, sizeof(unsigned *));
	unsigned *pn = (unsigned *)calloc(tm->nv, sizeof(unsigned)), *pc = (unsigned *)calloc(tm->nv, sizeof(unsigned));
	tm->edges = (unsigned *)malloc((size_t)tm->nf * 3 * 4 * sizeof(unsigned));
	tm->ne = 0;
	for (f = 0; f < tm->nf; f++)
	{
		const unsigned *v = &tm->faces[3 * f];
		for (j = 0; j < 3; j++)
		{
			unsigned a = v[j], b = v[(j + 1) % 3], edge = 0xffffffffu;
			for (k = 0; k < pn[a]; k++)
			{
				const unsigned *e = &tm->edges[4 * pe[a][k]];
				if ((e[0] == a || e[0] == b) && (e[1] == a || e[1] == b)) { edge = pe[a][k]; break; }
			}
			if (edge == 0xffffffffu)
			{
				unsigned *e = &tm->edges[4 * tm->ne];
				e[0] = a; e[1] = b; e[2] = f; e[3] = 0xffffffffu;
				edge = tm->ne++;
			}
			else
				tm->edges[4 * edge + 3] = f;
			{
				unsigned q[2]; int t; q[0] = a; q[1] = b;
				for (t = 0; t < 2; t++)
				{
					unsigned p = q[t];
					if (pn[p] == pc[p]) { pc[p] = pc[p] ? pc[p] * 2 : 8; pe[p] = (unsigned *)realloc(pe[p], pc[p] * sizeof(unsigned)); }
					pe[p][pn[p]++] = edge;
				}
			}
		}
	}
	for (k = 0; k < tm->nv; k++) free(pe[k]);
	free(pe); free(pn); free(pc);
}

/* IndexedTetMesh::buildNeighbors  Utils/IndexedTetMesh.cpp:55-182 (edges {01,02,03,12,13,23}) */
static void PO(build_tet_edges)(PO(tetmesh) *tm)
{
	static const int E[6][2] = { { 0, 1 }, { 0, 2 }, { 0, 3 }, { 1, 2 }, { 1, 3 }, { 2, 3 } };
	unsigned t, j, k;
	unsigned **ve = (unsigned **)calloc(tm->nv, sizeof(unsigned *));
	unsigned *vn = (unsigned *)calloc(tm->nv, sizeof(unsigned)), *vc = (unsigned *)calloc(tm->nv, sizeof(unsigned));
	tm->edges = (unsigned *)malloc((size_t)tm->nt * 6 * 2 * sizeof(unsigned));
	tm->vtets = (unsigned *)calloc(tm->nv, sizeof(unsigned));
	tm->ne = 0;
	for (t = 0; t < tm->nt; t++)
	{
		const unsigned *v = &tm->tets[4 * t];
		for (j = 0; j < 4; j++) tm->vtets[v[j]]++;
		for (j = 0; j < 6; j++)
		{
			unsigned a = v[E[j][0]], b = v[E[j][1]], edge = 0xffffffffu;
			for (k = 0; k < vn[a]; k++)
			{
				const unsigned *e = &tm->edges[2 * ve[a][k]];
				if ((e[0] == a || e[0] == b) && (e[1] == a || e[1] == b)) { edge = ve[a][k]; break; }
			}
			if (edge == 0xffffffffu)
			{
				unsigned q[2]; int u;
				tm->edges[2 * tm->ne] = a; tm->edges[2 * tm->ne + 1] = b;
				edge = tm->ne++;
				q[0] = a; q[1] = b;
				for (u = 0; u < 2; u++)
				{
					unsigned p = q[u];
					if (vn[p] == vc[p]) { vc[p] = vc[p] ? vc[p] * 2 : 16; ve[p] = (unsigned *)realloc(ve[p], vc[p] * sizeof(unsigned)); }
					ve[p][vn[p]++] = edge;
				}
			}
		}
	}
	for (k = 0; k < tm->nv; k++) free(ve[k]);
	free(ve); free(vn); free(vc);
}

/* SimulationModel::addTriangleModel  SimulationModel.cpp:806-829 */
int PO(add_triangle_model)(PO(sim) *s, unsigned np, unsigned nf, const double *pts, const unsigned *idx)
{
	PO(trimesh) *tm = &s->tri[s->ntri];
	unsigned i;
	tm->off = s->n; tm->nv = np; tm->nf = nf;
	for (i = 0; i < np; i++) PO(add_vertex)(s, pts + 3 * i);
	tm->faces = (unsigned *)malloc((size_t)nf * 3 * sizeof(unsigned));
	memcpy(tm->faces, idx, (size_t)nf * 3 * sizeof(unsigned));
	PO(build_tri_edges)(tm);
	return (int)s->ntri++;
}

/* SimulationModel::addRegularTriangleModel  SimulationModel.cpp:831-901; Rm row-major */
int PO(add_regular_triangle_model)(PO(sim) *s, int width, int height, const double *T, const double *Rm, const double *scale)
{
	M3 rot; int i, j, r, c, res; size_t index = 0;
	V3 t = MK((R)T[0], (R)T[1], (R)T[2]);
	R dy = (R)scale[1] / (R)(height - 1);
	R dx = (R)scale[0] / (R)(width - 1);
	double *pts = (double *)malloc((size_t)width * height * 3 * sizeof(double));
	unsigned *idx = (unsigned *)malloc((size_t)6 * (height - 1) * (width - 1) * sizeof(unsigned));
	for (r = 0; r < 3; r++) for (c = 0; c < 3; c++) rot.m[r][c] = (R)Rm[3 * r + c];
	for (i = 0; i < height; i++)
		for (j = 0; j < width; j++)
		{
			R y = dy * i, x = dx * j;
			V3 p = ADD(PO(mvec)(&rot, MK(x, y, 0)), t);
			size_t k = (size_t)i * width + j;
			pts[3 * k] = p.x; pts[3 * k + 1] = p.y; pts[3 * k + 2] = p.z;
		}
	for (i = 0; i < height - 1; i++)
		for (j = 0; j < width - 1; j++)
		{
			int helper = (i % 2 == j % 2) ? 1 : 0;
			idx[index] = i * width + j; idx[index + 1] = i * width + j + 1; idx[index + 2] = (i + 1) * width + j + helper;
			index += 3;
			idx[index] = (i + 1) * width + j + 1; idx[index + 1] = (i + 1) * width + j; idx[index + 2] = i * width + j + 1 - helper;
			index += 3;
		}
	res = PO(add_triangle_model)(s, (unsigned)(width * height), (unsigned)(index / 3), pts, idx);
	for (i = 0; i < width * height; i++) PO(set_mass)(s, s->tri[res].off + i, 1.0);
	free(pts); free(idx);
	return res;
}

/* SimulationModel::addTetModel  SimulationModel.cpp:903-919 */
int PO(add_tet_model)(PO(sim) *s, unsigned np, unsigned nt, const double *pts, const unsigned *idx)
{
	PO(tetmesh) *tm = &s->tet[s->ntet];
	unsigned i;
	tm->off = s->n; tm->nv = np; tm->nt = nt;
	for (i = 0; i < np; i++) PO(add_vertex)(s, pts + 3 * i);
	tm->tets = (unsigned *)malloc((size_t)nt * 4 * sizeof(unsigned));
	memcpy(tm->tets, idx, (size_t)nt * 4 * sizeof(unsigned));
	PO(build_tet_edges)(tm);
	return (int)s->ntet++;
}

/* SimulationModel::addRegularTetModel  SimulationModel.cpp:921-1005 */
int PO(add_regular_tet_model)(PO(sim) *s, int width, int height, int depth, const double *T, const double *Rm, const double *scale)
{
	M3 rot; int i, j, k, r, c, res; size_t n = 0;
	R dx = (R)scale[0] / (R)(width - 1), dy = (R)scale[1] / (R)(height - 1), dz = (R)scale[2] / (R)(depth - 1);
	V3 t = MK((R)T[0] - (R)0.5 * (R)scale[0], (R)T[1] - (R)0.5 * (R)scale[1], (R)T[2] - (R)0.5 * (R)scale[2]);
	double *pts = (double *)malloc((size_t)width * height * depth * 3 * sizeof(double));
	unsigned *idx = (unsigned *)malloc((size_t)(width - 1) * (height - 1) * (depth - 1) * 20 * sizeof(unsigned));
	for (r = 0; r < 3; r++) for (c = 0; c < 3; c++) rot.m[r][c] = (R)Rm[3 * r + c];
	for (i = 0; i < width; i++) for (j = 0; j < height; j++) for (k = 0; k < depth; k++)
	{
		V3 p = ADD(PO(mvec)(&rot, MK(dx * i, dy * j, dz * k)), t);
		size_t q = (size_t)i * height * depth + (size_t)j * depth + k;
		pts[3 * q] = p.x; pts[3 * q + 1] = p.y; pts[3 * q + 2] = p.z;
	}
	for (i = 0; i < width - 1; i++) for (j = 0; j < height - 1; j++) for (k = 0; k < depth - 1; k++)
	{
		unsigned p0 = i * height * depth + j * depth + k, p1 = p0 + 1;
		unsigned p3 = (i + 1) * height * depth + j * depth + k, p2 = p3 + 1;
		unsigned p7 = (i + 1) * height * depth + (j + 1) * depth + k, p6 = p7 + 1;
		unsigned p4 = i * height * depth + (j + 1) * depth + k, p5 = p4 + 1;
		if ((i + j + k) % 2 == 1)
		{
			unsigned q[20] = { p2, p1, p6, p3,  p6, p3, p4, p7,  p4, p1, p6, p5,  p3, p1, p4, p0,  p6, p1, p4, p3 };
			memcpy(idx + n, q, sizeof(q));
		}
		else
		{
			unsigned q[20] = { p0, p2, p5, p1,  p7, p2, p0, p3,  p5, p2, p7, p6,  p7, p0, p5, p4,  p0, p2, p7, p5 };
			memcpy(idx + n, q, sizeof(q));
		}
		n += 20;
	}
	res = PO(add_tet_model)(s, (unsigned)(width * height * depth), (unsigned)(n / 4), pts, idx);
	for (i = 0; i < width * height * depth; i++) PO(set_mass)(s, s->tet[res].off + i, 1.0);
	free(pts); free(idx);
	return res;
}

/* ------------------------------------------------------------------------- */
/* constraint initialisation (Simulation/Constraints.cpp *::initConstraint + init_* solvers) */
static PO(con) *PO(new_con)(PO(sim) *s, int type, const unsigned *b, unsigned nb)
{
	PO(con) *c;
	if (s->nc == s->ccap) { s->ccap = s->ccap ? s->ccap * 2 : 1024; s->c = (PO(con) *)realloc(s->c, s->ccap * sizeof(PO(con))); }
	c = &s->c[s->nc];
	memset(c, 0, sizeof(*c));
	c->type = type;
	memcpy(c->b, b, nb * sizeof(unsigned));
	return c;
}
static void PO(commit)(PO(sim) *s) { s->nc++; s->groups_ok = 0; }

static R PO(cot_theta)(V3 v, V3 w) { return DOT(v, w) / NRM(CRS(v, w)); }  /* MathFunctions.cpp:391-396 */

static const unsigned PO(nbodies)[13] = { 2, 2, 4, 4, 4, 3, 3, 4, 4, 4, 4, 4, 4 };
static const unsigned PO(nparams)[13] = { 2, 2, 2, 17, 17, 10, 9, 2, 2, 12, 12, 13, 24 };

static void PO(store_cm)(const M3 *A, R *o) { int c, r; for (c = 0; c < 3; c++) for (r = 0; r < 3; r++) o[c * 3 + r] = A->m[r][c]; }
static M3 PO(load_cm)(const R *o) { M3 A; int c, r; for (c = 0; c < 3; c++) for (r = 0; r < 3; r++) A.m[r][c] = o[c * 3 + r]; return A; }

/* generic add: `args` per type
 *  0,1,2,3,4,7,8: [stiffness]          5: [xx,yy,xy,xyPoisson,yxPoisson]   6: [xx,yy,xy,normStretch,normShear]
 *  9,10: [stiffness,poisson]           11: [stretch,shear,normStretch,normShear]   12: [stiffness] + nclusters[4]
 * returns 1 if the reference's initConstraint returns true */
int PO(add_constraint)(PO(sim) *s, int type, const unsigned *b, const double *args, const unsigned *nclusters)
{
	PO(con) *c = PO(new_con)(s, type, b, PO(nbodies)[type]);
	const R *X = s->x0;
	int i;
	switch (type)
	{
	case 0: case 1: /* Constraints.cpp:1166-1179, 1211-1225 */
		c->p[0] = NRM(SUB(PO(ld)(X, b[1]), PO(ld)(X, b[0])));
		c->p[1] = (R)args[0];
		break;
	case 2: /* Constraints.cpp:1264-1299 */
	{
		V3 p0 = PO(ld)(X, b[0]), p1 = PO(ld)(X, b[1]), p2 = PO(ld)(X, b[2]), p3 = PO(ld)(X, b[3]);
		V3 e = SUB(p3, p2), n1, n2;
		R elen = NRM(e), d;
		if ((double)elen < 1e-6) return 0;
		n1 = CRS(SUB(p2, p0), SUB(p3, p0)); n1 = DVS(n1, SQN(n1));
		n2 = CRS(SUB(p3, p1), SUB(p2, p1)); n2 = DVS(n2, SQN(n2));
		n1 = NMZ(n1); n2 = NMZ(n2);
		d = DOT(n1, n2);
		if (d < (R)-1) d = (R)-1;
		if (d > (R)1) d = (R)1;
		c->p[0] = PO_ACOS(d);
		c->p[1] = (R)args[0];
		break;
	}
	case 3: case 4: /* init_IsometricBendingConstraint  PositionBasedDynamics.cpp:145-183 */
	{
		V3 x[4], e0, e1, e2, e3, e4;
		R c01, c02, c03, c04, A0, A1, coef, K[4], K2[4];
		int j, k;
		x[0] = PO(ld)(X, b[2]); x[1] = PO(ld)(X, b[3]); x[2] = PO(ld)(X, b[0]); x[3] = PO(ld)(X, b[1]);
		e0 = SUB(x[1], x[0]); e1 = SUB(x[2], x[0]); e2 = SUB(x[3], x[0]); e3 = SUB(x[2], x[1]); e4 = SUB(x[3], x[1]);
		c01 = PO(cot_theta)(e0, e1); c02 = PO(cot_theta)(e0, e2);
		c03 = PO(cot_theta)(NEG(e0), e3); c04 = PO(cot_theta)(NEG(e0), e4);
		A0 = (R)0.5 * NRM(CRS(e0, e1)); A1 = (R)0.5 * NRM(CRS(e0, e2));
		coef = (R)(-3.f) / ((R)2.f * (A0 + A1));
		K[0] = c03 + c04; K[1] = c01 + c02; K[2] = -c01 - c03; K[3] = -c02 - c04;
		for (j = 0; j < 4; j++) K2[j] = coef * K[j];
		c->p[0] = (R)args[0];
		for (j = 0; j < 4; j++)
		{
			for (k = 0; k < j; k++) c->p[1 + k * 4 + j] = c->p[1 + j * 4 + k] = K[j] * K2[k];
			c->p[1 + j * 4 + j] = K[j] * K2[j];
		}
		break;
	}
	case 5: /* init_FEMTriangleConstraint  PositionBasedDynamics.cpp:808-841 */
	{
		V3 p0 = PO(ld)(X, b[0]), p1 = PO(ld)(X, b[1]), p2 = PO(ld)(X, b[2]);
		V3 n0 = CRS(SUB(p1, p0), SUB(p2, p0));
		V3 a1 = NMZ(SUB(p1, p0));
		V3 a2 = NMZ(CRS(n0, a1));
		R q[3][2], P00, P10, P01, P11, det, id;
		q[0][0] = DOT(p0, a2); q[0][1] = DOT(p0, a1); q[1][0] = DOT(p1, a2); q[1][1] = DOT(p1, a1); q[2][0] = DOT(p2, a2); q[2][1] = DOT(p2, a1);
		P00 = q[0][0] - q[2][0]; P10 = q[0][1] - q[2][1]; P01 = q[1][0] - q[2][0]; P11 = q[1][1] - q[2][1];
		det = P00 * P11 - P10 * P01;
		if (!(PO_FABS(det) > EPS)) return 0;
		id = (R)1 / det;
		c->p[0] = NRM(n0) * (R)0.5;
		c->p[1] = P11 * id; c->p[2] = -P10 * id; c->p[3] = -P01 * id; c->p[4] = P00 * id;
		for (i = 0; i < 5; i++) c->p[5 + i] = (R)args[i];
		break;
	}
	case 6: /* Constraints.cpp:1544-1569 (x-z plane) + init_StrainTriangleConstraint PositionBasedDynamics.cpp:562-581 */
	{
		V3 x1 = PO(ld)(X, b[0]), x2 = PO(ld)(X, b[1]), x3 = PO(ld)(X, b[2]);
		R a = x2.x - x1.x, bb = x3.x - x1.x, cc = x2.z - x1.z, d = x3.z - x1.z;
		R det = a * d - bb * cc, sc;
		if (PO_FABS(det) < EPS) return 0;
		sc = (R)1 / det;
		c->p[0] = d * sc; c->p[1] = -cc * sc; c->p[2] = -bb * sc; c->p[3] = a * sc;
		c->p[4] = (R)args[0]; c->p[5] = (R)args[1]; c->p[6] = (R)args[2]; c->p[7] = args[3] != 0; c->p[8] = args[4] != 0;
		break;
	}
	case 7: case 8: case 9: case 10:
	{
		V3 p0 = PO(ld)(X, b[0]), p1 = PO(ld)(X, b[1]), p2 = PO(ld)(X, b[2]), p3 = PO(ld)(X, b[3]);
		/* Constraints.cpp:1632 / PositionBasedDynamics.cpp:941 */
		R vol = PO_FABS((R)(1.0 / 6.0) * DOT(SUB(p3, p0), CRS(SUB(p2, p0), SUB(p1, p0))));
		c->p[0] = vol;
		if (type <= 8) { c->p[1] = (R)args[0]; break; }
		{
			M3 m, inv; R det;   /* init_FEMTetraConstraint  PositionBasedDynamics.cpp:933-955 */
			PO(setcol)(&m, 0, SUB(p0, p3)); PO(setcol)(&m, 1, SUB(p1, p3)); PO(setcol)(&m, 2, SUB(p2, p3));
			det = PO(mdet)(&m);
			if (!(PO_FABS(det) > EPS)) return 0;
			inv = PO(minv)(&m);
			PO(store_cm)(&inv, &c->p[1]);
			c->p[10] = (R)args[0]; c->p[11] = (R)args[1];
		}
		break;
	}
	case 11: /* init_StrainTetraConstraint  PositionBasedDynamics.cpp:691-710 */
	{
		V3 p0 = PO(ld)(X, b[0]), p1 = PO(ld)(X, b[1]), p2 = PO(ld)(X, b[2]), p3 = PO(ld)(X, b[3]);
		M3 m, inv; R det;
		PO(setcol)(&m, 0, SUB(p1, p0)); PO(setcol)(&m, 1, SUB(p2, p0)); PO(setcol)(&m, 2, SUB(p3, p0));
		det = PO(mdet)(&m);
		if (!(PO_FABS(det) > EPS)) return 0;
		inv = PO(minv)(&m);
		PO(store_cm)(&inv, &c->p[0]);
		c->p[9] = (R)args[0]; c->p[10] = (R)args[1]; c->p[11] = args[2] != 0; c->p[12] = args[3] != 0;
		break;
	}
	case 12: /* Constraints.cpp:1985-2001 + init_ShapeMatchingConstraint PositionBasedDynamics.cpp:481-498 */
	{
		V3 cm = MK(0, 0, 0); R wsum = 0;
		c->p[0] = (R)args[0];
		for (i = 0; i < 4; i++)
		{
			V3 x0 = PO(ld)(X, b[i]);
			R w = s->w[b[i]], wi = (R)1 / (w + EPS);
			c->p[4 + 3 * i] = x0.x; c->p[5 + 3 * i] = x0.y; c->p[6 + 3 * i] = x0.z;
			c->p[16 + i] = w; c->p[20 + i] = (R)nclusters[i];
			cm = ADD(cm, SCR(x0, wi)); wsum += wi;
		}
		if (wsum == 0) return 0;
		cm = DVS(cm, wsum);
		c->p[1] = cm.x; c->p[2] = cm.y; c->p[3] = cm.z;
		break;
	}
	default: return 0;
	}
	PO(commit)(s);
	return 1;
}

/* SimulationModel::addClothConstraints  SimulationModel.cpp:1125-1184 */
void PO(add_cloth_constraints)(PO(sim) *s, unsigned tmi, unsigned method, double k, double xx, double yy, double xy, double xyP, double yxP, int ns, int nsh)
{
	PO(trimesh) *tm = &s->tri[tmi];
	unsigned i;
	if (method == 1 || method == 4)
		for (i = 0; i < tm->ne; i++)
		{
			unsigned b[2]; double a[1];
			b[0] = tm->edges[4 * i] + tm->off; b[1] = tm->edges[4 * i + 1] + tm->off; a[0] = k;
			PO(add_constraint)(s, method == 1 ? 0 : 1, b, a, NULL);
		}
	else if (method == 2 || method == 3)
		for (i = 0; i < tm->nf; i++)
		{
			unsigned b[3]; double a[5];
			b[0] = tm->faces[3 * i] + tm->off; b[1] = tm->faces[3 * i + 1] + tm->off; b[2] = tm->faces[3 * i + 2] + tm->off;
			a[0] = xx; a[1] = yy; a[2] = xy;
			if (method == 2) { a[3] = xyP; a[4] = yxP; PO(add_constraint)(s, 5, b, a, NULL); }
			else { a[3] = ns; a[4] = nsh; PO(add_constraint)(s, 6, b, a, NULL); }
		}
}

/* SimulationModel::addBendingConstraints  SimulationModel.cpp:1186-1240 */
void PO(add_bending_constraints)(PO(sim) *s, unsigned tmi, unsigned method, double k)
{
	PO(trimesh) *tm = &s->tri[tmi];
	unsigned i; int j;
	if (method < 1 || method > 3) return;
	for (i = 0; i < tm->ne; i++)
	{
		const unsigned *e = &tm->edges[4 * i];
		int p1 = -1, p2 = -1;
		if (e[2] == 0xffffffffu || e[3] == 0xffffffffu) continue;
		for (j = 0; j < 3; j++) if (tm->faces[3 * e[2] + j] != e[0] && tm->faces[3 * e[2] + j] != e[1]) { p1 = (int)tm->faces[3 * e[2] + j]; break; }
		for (j = 0; j < 3; j++) if (tm->faces[3 * e[3] + j] != e[0] && tm->faces[3 * e[3] + j] != e[1]) { p2 = (int)tm->faces[3 * e[3] + j]; break; }
		if (p1 != -1 && p2 != -1)
		{
			unsigned b[4]; double a[1];
			b[0] = p1 + tm->off; b[1] = p2 + tm->off; b[2] = e[0] + tm->off; b[3] = e[1] + tm->off; a[0] = k;
			PO(add_constraint)(s, method == 1 ? 2 : (method == 2 ? 3 : 4), b, a, NULL);
		}
	}
}

/* SimulationModel::addSolidConstraints  SimulationModel.cpp:1242-1349 */
void PO(add_solid_constraints)(PO(sim) *s, unsigned tmi, unsigned method, double k, double nu, double kv, int ns, int nsh)
{
	PO(tetmesh) *tm = &s->tet[tmi];
	unsigned i; int j;
	(void)nsh; /* the reference passes normalizeStretch twice, SimulationModel.cpp:1308 */
	if (method == 1 || method == 6)
		for (i = 0; i < tm->ne; i++)
		{
			unsigned b[2]; double a[1];
			b[0] = tm->edges[2 * i] + tm->off; b[1] = tm->edges[2 * i + 1] + tm->off; a[0] = k;
			PO(add_constraint)(s, method == 1 ? 0 : 1, b, a, NULL);
		}
	for (i = 0; i < tm->nt; i++)
	{
		unsigned b[4], nc[4]; double a[4];
		for (j = 0; j < 4; j++) b[j] = tm->tets[4 * i + j] + tm->off;
		switch (method)
		{
		case 1: a[0] = kv; PO(add_constraint)(s, 7, b, a, NULL); break;
		case 2: a[0] = k; a[1] = nu; PO(add_constraint)(s, 9, b, a, NULL); break;
		case 3: a[0] = k; a[1] = nu; PO(add_constraint)(s, 10, b, a, NULL); break;
		case 4: a[0] = k; a[1] = k; a[2] = ns; a[3] = ns; PO(add_constraint)(s, 11, b, a, NULL); break;
		case 5: for (j = 0; j < 4; j++) nc[j] = tm->vtets[b[j] - tm->off]; a[0] = k; PO(add_constraint)(s, 12, b, a, nc); break;
		case 6: a[0] = kv; PO(add_constraint)(s, 8, b, a, NULL); break;
		default: break;
		}
	}
}

/* SimulationModel::initConstraintGroups  SimulationModel.cpp:1033-1094 (byte map per group, first fit) */
void PO(init_constraint_groups)(PO(sim) *s)
{
	unsigned i, j, k; unsigned char **map = NULL; unsigned *gcap = NULL;
	if (s->groups_ok) return;
	PO(free_groups)(s);
	for (i = 0; i < s->nc; i++)
	{
		const PO(con) *c = &s->c[i];
		unsigned nb = PO(nbodies)[c->type];
		int placed = 0;
		for (j = 0; j < s->ng && !placed; j++)
		{
			int ok = 1;
			for (k = 0; k < nb; k++) if (map[j][c->b[k]]) { ok = 0; break; }
			if (ok)
			{
				if (s->gsz[j] == gcap[j]) { gcap[j] *= 2; s->grp[j] = (unsigned *)realloc(s->grp[j], gcap[j] * sizeof(unsigned)); }
				s->grp[j][s->gsz[j]++] = i;
				for (k = 0; k < nb; k++) map[j][c->b[k]] = 1;
				placed = 1;
			}
		}
		if (!placed)
		{
			s->ng++;
			map = (unsigned char **)realloc(map, s->ng * sizeof(*map));
			s->grp = (unsigned **)realloc(s->grp, s->ng * sizeof(*s->grp));
			s->gsz = (unsigned *)realloc(s->gsz, s->ng * sizeof(unsigned));
			gcap = (unsigned *)realloc(gcap, s->ng * sizeof(unsigned));
			map[s->ng - 1] = (unsigned char *)calloc(s->n ? s->n : 1, 1);
			gcap[s->ng - 1] = 256; s->gsz[s->ng - 1] = 0;
			s->grp[s->ng - 1] = (unsigned *)malloc(256 * sizeof(unsigned));
			s->grp[s->ng - 1][s->gsz[s->ng - 1]++] = i;
			for (k = 0; k < nb; k++) map[s->ng - 1][c->b[k]] = 1;
		}
	}
	for (j = 0; j < s->ng; j++) free(map[j]);
	free(map); free(gcap);
	s->groups_ok = 1;
}

/* ------------------------------------------------------------------------- */
/* projections: each returns 1 when the reference applies the corrections */

/* PositionBasedDynamics.cpp:13-34 */
static int PO(solve_distance)(V3 p0, R w0, V3 p1, R w1, R L, R k, V3 *c0, V3 *c1)
{
	R wSum = w0 + w1, d, dl; V3 n, corr;
	if (wSum == 0) return 0;
	n = SUB(p1, p0); d = NRM(n); n = NMZ(n); dl = d - L;
	corr = MK(((k * n.x) * dl) / wSum, ((k * n.y) * dl) / wSum, ((k * n.z) * dl) / wSum);
	*c0 = SCL(w0, corr); *c1 = SCL(-w1, corr);
	return 1;
}

/* XPBD.cpp:14-60 */
static int PO(solve_distance_xpbd)(V3 p0, R w0, V3 p1, R w1, R L, R k, R dt, R *lambda, V3 *c0, V3 *c1)
{
	R K = w0 + w1, d, C, alpha = 0, Kinv, dl; V3 n = SUB(p0, p1), pt;
	d = NRM(n); C = d - L;
	*c0 = MK(0, 0, 0); *c1 = *c0;
	if (d > (R)1e-6) n = DVS(n, d); else return 1;
	if (k != 0) { alpha = (R)1 / (k * dt * dt); K += alpha; }
	if (PO_FABS(K) > (R)1e-6) Kinv = (R)1 / K; else return 1;
	dl = -Kinv * (C + alpha * *lambda);
	*lambda += dl;
	pt = SCR(n, dl);
	*c0 = SCL(w0, pt); *c1 = SCL(-w1, pt);
	return 1;
}

/* PositionBasedDynamics.cpp:37-102 */
static int PO(solve_dihedral)(const V3 *p, const R *w, R rest, R k, V3 *c)
{
	V3 e, n1, n2, d0, d1, d2, d3; R elen, inv, dt, phi, lambda;
	if (w[0] == 0 && w[1] == 0) return 0;
	e = SUB(p[3], p[2]); elen = NRM(e);
	if (elen < EPS) return 0;
	inv = (R)1 / elen;
	n1 = CRS(SUB(p[2], p[0]), SUB(p[3], p[0])); n1 = DVS(n1, SQN(n1));
	n2 = CRS(SUB(p[3], p[1]), SUB(p[2], p[1])); n2 = DVS(n2, SQN(n2));
	d0 = SCL(elen, n1); d1 = SCL(elen, n2);
	d2 = ADD(SCL(DOT(SUB(p[0], p[3]), e) * inv, n1), SCL(DOT(SUB(p[1], p[3]), e) * inv, n2));
	d3 = ADD(SCL(DOT(SUB(p[2], p[0]), e) * inv, n1), SCL(DOT(SUB(p[2], p[1]), e) * inv, n2));
	n1 = NMZ(n1); n2 = NMZ(n2);
	dt = DOT(n1, n2);
	if (dt < (R)-1) dt = (R)-1;
	if (dt > (R)1) dt = (R)1;
	phi = (R)acos((double)dt);   /* PositionBasedDynamics.cpp:73 resolves to ::acos(double): no <math.h> C++ overloads in that TU */
	lambda = w[0] * SQN(d0) + w[1] * SQN(d1) + w[2] * SQN(d2) + w[3] * SQN(d3);
	if (lambda == 0) return 0;
	lambda = (phi - rest) / lambda * k;
	if (DOT(CRS(n1, n2), e) > 0) lambda = -lambda;
	c[0] = SCL(-w[0] * lambda, d0); c[1] = SCL(-w[1] * lambda, d1); c[2] = SCL(-w[2] * lambda, d2); c[3] = SCL(-w[3] * lambda, d3);
	return 1;
}

/* PositionBasedDynamics.cpp:104-142 / XPBD.cpp:63-109 */
static int PO(solve_volume)(const V3 *p, const R *w, R rest, R k, int xpbd, R dt, R *lam, V3 *c)
{
	R volume = (R)(1.0 / 6.0) * DOT(CRS(SUB(p[1], p[0]), SUB(p[2], p[0])), SUB(p[3], p[0]));
	V3 g0 = CRS(SUB(p[1], p[2]), SUB(p[3], p[2]));
	V3 g1 = CRS(SUB(p[2], p[0]), SUB(p[3], p[0]));
	V3 g2 = CRS(SUB(p[0], p[1]), SUB(p[3], p[1]));
	V3 g3 = CRS(SUB(p[1], p[0]), SUB(p[2], p[0]));
	R K;
	c[0] = MK(0, 0, 0); c[1] = c[0]; c[2] = c[0]; c[3] = c[0];
	if (!xpbd)
	{
		R lambda;
		if (k == 0) return 0;
		lambda = w[0] * SQN(g0) + w[1] * SQN(g1) + w[2] * SQN(g2) + w[3] * SQN(g3);
		if (PO_FABS(lambda) < EPS) return 0;
		lambda = k * (volume - rest) / lambda;
		c[0] = SCL(-lambda * w[0], g0); c[1] = SCL(-lambda * w[1], g1); c[2] = SCL(-lambda * w[2], g2); c[3] = SCL(-lambda * w[3], g3);
		return 1;
	}
	K = w[0] * SQN(g0) + w[1] * SQN(g1) + w[2] * SQN(g2) + w[3] * SQN(g3);
	{
		R alpha = 0, C, dl;
		if (k != 0) { alpha = (R)1 / (k * dt * dt); K += alpha; }
		if (PO_FABS(K) < EPS) return 0;
		C = volume - rest;
		dl = -(C + alpha * *lam) / K;
		*lam += dl;
		c[0] = SCL(dl * w[0], g0); c[1] = SCL(dl * w[1], g1); c[2] = SCL(dl * w[2], g2); c[3] = SCL(dl * w[3], g3);
	}
	return 1;
}

/* PositionBasedDynamics.cpp:186-236 / XPBD.cpp:153-213; Q column-major Q(j,k) = q[k*4+j] */
static int PO(solve_isometric)(const V3 *p, const R *wi, const R *q, R k, int xpbd, R dt, R *lam, V3 *c)
{
	V3 x[4], g[4]; R w[4], energy = 0, sum = 0; int j, kk;
	x[0] = p[2]; x[1] = p[3]; x[2] = p[0]; x[3] = p[1];
	w[0] = wi[2]; w[1] = wi[3]; w[2] = wi[0]; w[3] = wi[1];
	for (kk = 0; kk < 4; kk++) for (j = 0; j < 4; j++) energy += q[kk * 4 + j] * DOT(x[kk], x[j]);
	energy *= (R)0.5;
	for (j = 0; j < 4; j++) g[j] = MK(0, 0, 0);
	for (kk = 0; kk < 4; kk++) for (j = 0; j < 4; j++) g[j] = ADD(g[j], SCL(q[kk * 4 + j], x[kk]));
	for (j = 0; j < 4; j++) if (w[j] != 0) sum += w[j] * SQN(g[j]);
	if (!xpbd)
	{
		if (PO_FABS(sum) > EPS)
		{
			R s = energy / sum;
			c[0] = SCL(-k * (s * w[2]), g[2]); c[1] = SCL(-k * (s * w[3]), g[3]); c[2] = SCL(-k * (s * w[0]), g[0]); c[3] = SCL(-k * (s * w[1]), g[1]);
			return 1;
		}
		return 0;
	}
	{
		R alpha = 0;
		if (k != 0) { alpha = (R)1 / (k * dt * dt); sum += alpha; }
		if (PO_FABS(sum) > EPS)
		{
			R dl = -(energy + alpha * *lam) / sum;
			*lam += dl;
			c[0] = SCL(dl * w[2], g[2]); c[1] = SCL(dl * w[3], g[3]); c[2] = SCL(dl * w[0], g[0]); c[3] = SCL(dl * w[1], g[1]);
			return 1;
		}
	}
	return 0;
}

/* PositionBasedDynamics.cpp:844-930; par = [area, im00, im10, im01, im11, xx, yy, xy, nuXY, nuYX] */
static int PO(solve_fem_triangle)(const V3 *p, const R *w, const R *par, V3 *c)
{
	R area = par[0], im[2][2], Ex = par[5], Ey = par[6], Es = par[7], nxy = par[8], nyx = par[9];
	R Cm[3][3], F[3][2], eps[2][2], st[2][2], PK[3][2], H[3][2], psi = 0, energy, sum;
	V3 p13 = SUB(p[0], p[2]), p23 = SUB(p[1], p[2]), g0, g1, g2; int i, j, k;
	im[0][0] = par[1]; im[1][0] = par[2]; im[0][1] = par[3]; im[1][1] = par[4];
	memset(Cm, 0, sizeof(Cm));
	Cm[0][0] = Ex / ((R)1 - nxy * nyx); Cm[0][1] = Ex * nyx / ((R)1 - nxy * nyx);
	Cm[1][1] = Ey / ((R)1 - nxy * nyx); Cm[1][0] = Ey * nxy / ((R)1 - nxy * nyx); Cm[2][2] = Es;
	F[0][0] = p13.x * im[0][0] + p23.x * im[1][0]; F[0][1] = p13.x * im[0][1] + p23.x * im[1][1];
	F[1][0] = p13.y * im[0][0] + p23.y * im[1][0]; F[1][1] = p13.y * im[0][1] + p23.y * im[1][1];
	F[2][0] = p13.z * im[0][0] + p23.z * im[1][0]; F[2][1] = p13.z * im[0][1] + p23.z * im[1][1];
	eps[0][0] = (R)0.5 * (F[0][0] * F[0][0] + F[1][0] * F[1][0] + F[2][0] * F[2][0] - (R)1);
	eps[1][1] = (R)0.5 * (F[0][1] * F[0][1] + F[1][1] * F[1][1] + F[2][1] * F[2][1] - (R)1);
	eps[0][1] = (R)0.5 * (F[0][0] * F[0][1] + F[1][0] * F[1][1] + F[2][0] * F[2][1]);
	eps[1][0] = eps[0][1];
	st[0][0] = Cm[0][0] * eps[0][0] + Cm[0][1] * eps[1][1] + Cm[0][2] * eps[0][1];
	st[1][1] = Cm[1][0] * eps[0][0] + Cm[1][1] * eps[1][1] + Cm[1][2] * eps[0][1];
	st[0][1] = Cm[2][0] * eps[0][0] + Cm[2][1] * eps[1][1] + Cm[2][2] * eps[0][1];
	st[1][0] = st[0][1];
	for (i = 0; i < 3; i++) for (j = 0; j < 2; j++) PK[i][j] = F[i][0] * st[0][j] + F[i][1] * st[1][j];
	for (j = 0; j < 2; j++) for (k = 0; k < 2; k++) psi += eps[j][k] * st[j][k];
	psi = (R)0.5 * psi; energy = area * psi;
	for (i = 0; i < 3; i++) for (j = 0; j < 2; j++) H[i][j] = (area * PK[i][0]) * im[j][0] + (area * PK[i][1]) * im[j][1];
	g0 = MK(H[0][0], H[1][0], H[2][0]); g1 = MK(H[0][1], H[1][1], H[2][1]); g2 = SUB(NEG(g0), g1);
	sum = w[0] * SQN(g0); sum += w[1] * SQN(g1); sum += w[2] * SQN(g2);
	if (PO_FABS(sum) > EPS)
	{
		R s = energy / sum;
		c[0] = SCL(-(s * w[0]), g0); c[1] = SCL(-(s * w[1]), g1); c[2] = SCL(-(s * w[2]), g2);
		return 1;
	}
	return 0;
}

/* PositionBasedDynamics.cpp:584-688; par = [im00, im10, im01, im11, xx, yy, xy, normStretch, normShear] */
static int PO(solve_strain_triangle)(const V3 *p, const R *w, const R *par, V3 *corr)
{
	R im[2][2]; V3 c[2], r[3]; int i, j, k;
	int nStretch = par[7] != 0, nShear = par[8] != 0;
	im[0][0] = par[0]; im[1][0] = par[1]; im[0][1] = par[2]; im[1][1] = par[3];
	c[0] = MK(im[0][0], im[1][0], 0); c[1] = MK(im[0][1], im[1][1], 0);
	corr[0] = MK(0, 0, 0); corr[1] = corr[0]; corr[2] = corr[0];
	for (i = 0; i < 2; i++) for (j = 0; j <= i; j++)
	{
		R Sij = 0, lambda; V3 d[3];
		r[0] = MK((p[1].x + corr[1].x) - (p[0].x + corr[0].x), (p[2].x + corr[2].x) - (p[0].x + corr[0].x), 0);
		r[1] = MK((p[1].y + corr[1].y) - (p[0].y + corr[0].y), (p[2].y + corr[2].y) - (p[0].y + corr[0].y), 0);
		r[2] = MK((p[1].z + corr[1].z) - (p[0].z + corr[0].z), (p[2].z + corr[2].z) - (p[0].z + corr[0].z), 0);
		for (k = 0; k < 3; k++) Sij += DOT(r[k], c[i]) * DOT(r[k], c[j]);
		d[0] = MK(0, 0, 0);
		for (k = 0; k < 2; k++)
		{
			d[k + 1] = SCR(MK(DOT(r[0], c[j]), DOT(r[1], c[j]), DOT(r[2], c[j])), im[k][i]);
			d[k + 1] = ADD(d[k + 1], SCR(MK(DOT(r[0], c[i]), DOT(r[1], c[i]), DOT(r[2], c[i])), im[k][j]));
			d[0] = SUB(d[0], d[k + 1]);
		}
		if (i != j && nShear)
		{
			R fi2 = 0, fj2 = 0, fi, fj, sc;
			for (k = 0; k < 3; k++) { fi2 += DOT(r[k], c[i]) * DOT(r[k], c[i]); fj2 += DOT(r[k], c[j]) * DOT(r[k], c[j]); }
			fi = PO_SQRT(fi2); fj = PO_SQRT(fj2);
			d[0] = MK(0, 0, 0);
			sc = Sij / (fi2 * fi * fj2 * fj);
			for (k = 0; k < 2; k++)
			{
				d[k + 1] = DVS(d[k + 1], fi * fj);
				d[k + 1] = SUB(d[k + 1], SCR(SCR(SCL(fj * fj, MK(DOT(r[0], c[i]), DOT(r[1], c[i]), DOT(r[2], c[i]))), im[k][i]), sc));
				d[k + 1] = SUB(d[k + 1], SCR(SCR(SCL(fi * fi, MK(DOT(r[0], c[j]), DOT(r[1], c[j]), DOT(r[2], c[j]))), im[k][j]), sc));
				d[0] = SUB(d[0], d[k + 1]);
			}
			Sij = Sij / (fi * fj);
		}
		lambda = w[0] * SQN(d[0]) + w[1] * SQN(d[1]) + w[2] * SQN(d[2]);
		if (lambda == 0) continue;
		if (i == j)
		{
			R kk = (i == 0) ? par[4] : par[5];
			if (nStretch) { R sq = PO_SQRT(Sij); lambda = (R)2 * sq * (sq - (R)1) / lambda * kk; }
			else lambda = (Sij - (R)1) / lambda * kk;
		}
		else
			lambda = Sij / lambda * par[6];
		corr[0] = SUB(corr[0], SCL(lambda * w[0], d[0]));
		corr[1] = SUB(corr[1], SCL(lambda * w[1], d[1]));
		corr[2] = SUB(corr[2], SCL(lambda * w[2], d[2]));
	}
	return 1;
}

/* MathFunctions.cpp:11-43 */
static void PO(jacobi_rotate)(M3 *A, M3 *Rm, int p, int q)
{
	R d, t, c, s; int k;
	if (A->m[p][q] == 0) return;
	d = (A->m[p][p] - A->m[q][q]) / ((R)2 * A->m[p][q]);
	/* MathFunctions.cpp:18,20: ::fabs / ::sqrt are the double functions in that TU, so the sum and the
	 * quotient are double expressions narrowed to Real on assignment */
	t = (R)((double)(R)1 / (fabs((double)d) + sqrt((double)(d * d + (R)1))));
	if (d < 0) t = -t;
	c = (R)((double)(R)1 / sqrt((double)(t * t + 1)));
	s = t * c;
	A->m[p][p] += t * A->m[p][q];
	A->m[q][q] -= t * A->m[p][q];
	A->m[p][q] = A->m[q][p] = 0;
	for (k = 0; k < 3; k++)
		if (k != p && k != q)
		{
			R Akp = c * A->m[k][p] + s * A->m[k][q];
			R Akq = -s * A->m[k][p] + c * A->m[k][q];
			A->m[k][p] = A->m[p][k] = Akp;
			A->m[k][q] = A->m[q][k] = Akq;
		}
	for (k = 0; k < 3; k++)
	{
		R Rkp = c * Rm->m[k][p] + s * Rm->m[k][q];
		R Rkq = -s * Rm->m[k][p] + c * Rm->m[k][q];
		Rm->m[k][p] = Rkp; Rm->m[k][q] = Rkq;
	}
}

/* MathFunctions.cpp:46-75 */
static void PO(eigen_decomposition)(const M3 *A, M3 *vecs, R *vals)
{
	const R epsilon = (R)1e-15;
	M3 D = *A; int iter = 0, i, j;
	for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) vecs->m[i][j] = (i == j);
	while (iter < 10)
	{
		int p = 0, q = 1; R a, mx = PO_FABS(D.m[0][1]);
		a = PO_FABS(D.m[0][2]); if (a > mx) { p = 0; q = 2; mx = a; }
		a = PO_FABS(D.m[1][2]); if (a > mx) { p = 1; q = 2; mx = a; }
		if (mx < epsilon) break;
		PO(jacobi_rotate)(&D, vecs, p, q);
		iter++;
	}
	vals[0] = D.m[0][0]; vals[1] = D.m[1][1]; vals[2] = D.m[2][2];
}

/* MathFunctions.cpp:261-388 */
static void PO(svd_inv)(const M3 *A, R *sigma, M3 *U, M3 *VT)
{
	M3 At = PO(mtr)(A), AT_A = PO(mmul)(&At, A), V; R S[3], detV, detU; int l, m, chk = 0, pos = 0;
	PO(eigen_decomposition)(&AT_A, &V, S);
	detV = PO(mdet)(&V);
	if (detV < 0)
	{
		R minL = PO_REAL_MAX; int ps = 0;
		for (l = 0; l < 3; l++) if (S[l] < minL) { ps = l; minL = S[l]; }
		V.m[0][ps] = -V.m[0][ps]; V.m[1][ps] = -V.m[1][ps]; V.m[2][ps] = -V.m[2][ps];
	}
	for (l = 0; l < 3; l++) if (S[l] < 0) S[l] = 0;
	for (l = 0; l < 3; l++) sigma[l] = PO_SQRT(S[l]);
	*VT = PO(mtr)(&V);
	for (l = 0; l < 3; l++) if ((double)PO_FABS(sigma[l]) < 1.0e-4) { pos = l; chk++; }
	if (chk > 0)
	{
		if (chk > 1) { for (l = 0; l < 3; l++) for (m = 0; m < 3; m++) U->m[l][m] = (l == m); }
		else
		{
			V3 v[2], vec; int index = 0;
			*U = PO(mmul)(A, &V);
			for (l = 0; l < 3; l++) if (l != pos) for (m = 0; m < 3; m++) U->m[m][l] *= (R)1 / sigma[l];
			for (l = 0; l < 3; l++) if (l != pos) v[index++] = PO(col)(U, l);
			vec = NMZ(CRS(v[0], v[1]));
			PO(setcol)(U, pos, vec);
		}
	}
	else
	{
		R si[3]; for (l = 0; l < 3; l++) si[l] = (R)1 / sigma[l];
		*U = PO(mmul)(A, &V);
		for (l = 0; l < 3; l++) for (m = 0; m < 3; m++) U->m[m][l] *= si[l];
	}
	detU = PO(mdet)(U);
	if (detU < 0)
	{
		R minL = PO_REAL_MAX; int ps = 0;
		for (l = 0; l < 3; l++) if (sigma[l] < minL) { ps = l; minL = sigma[l]; }
		sigma[ps] = -sigma[ps];
		U->m[0][ps] = -U->m[0][ps]; U->m[1][ps] = -U->m[1][ps]; U->m[2][ps] = -U->m[2][ps];
	}
}

/* F written out element-wise  PositionBasedDynamics.cpp:965-979 */
static M3 PO(defgrad)(const V3 *x, const M3 *im)
{
	V3 a = SUB(x[0], x[3]), b = SUB(x[1], x[3]), c = SUB(x[2], x[3]); M3 F; int k;
	for (k = 0; k < 3; k++)
	{
		F.m[0][k] = a.x * im->m[0][k] + b.x * im->m[1][k] + c.x * im->m[2][k];
		F.m[1][k] = a.y * im->m[0][k] + b.y * im->m[1][k] + c.y * im->m[2][k];
		F.m[2][k] = a.z * im->m[0][k] + b.z * im->m[1][k] + c.z * im->m[2][k];
	}
	return F;
}

/* computeGreenStrainAndPiolaStress  PositionBasedDynamics.cpp:958-1008 */
static void PO(green_piola)(const V3 *x, const M3 *im, R V0, R mu, R lambda, M3 *sigma, R *energy)
{
	M3 F = PO(defgrad)(x, im), e, s; R trace, ltrace, psi = 0; int i, j;
	e.m[0][0] = (R)0.5 * (F.m[0][0] * F.m[0][0] + F.m[1][0] * F.m[1][0] + F.m[2][0] * F.m[2][0] - (R)1);
	e.m[1][1] = (R)0.5 * (F.m[0][1] * F.m[0][1] + F.m[1][1] * F.m[1][1] + F.m[2][1] * F.m[2][1] - (R)1);
	e.m[2][2] = (R)0.5 * (F.m[0][2] * F.m[0][2] + F.m[1][2] * F.m[1][2] + F.m[2][2] * F.m[2][2] - (R)1);
	e.m[0][1] = (R)0.5 * (F.m[0][0] * F.m[0][1] + F.m[1][0] * F.m[1][1] + F.m[2][0] * F.m[2][1]);
	e.m[0][2] = (R)0.5 * (F.m[0][0] * F.m[0][2] + F.m[1][0] * F.m[1][2] + F.m[2][0] * F.m[2][2]);
	e.m[1][2] = (R)0.5 * (F.m[0][1] * F.m[0][2] + F.m[1][1] * F.m[1][2] + F.m[2][1] * F.m[2][2]);
	e.m[1][0] = e.m[0][1]; e.m[2][0] = e.m[0][2]; e.m[2][1] = e.m[1][2];
	trace = e.m[0][0] + e.m[1][1] + e.m[2][2];
	ltrace = lambda * trace;
	for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) s.m[i][j] = e.m[i][j] * (R)2.0 * mu;
	s.m[0][0] += ltrace; s.m[1][1] += ltrace; s.m[2][2] += ltrace;
	*sigma = PO(mmul)(&F, &s);
	for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) psi += e.m[i][j] * e.m[i][j];
	psi = mu * psi + (R)0.5 * lambda * trace * trace;
	*energy = V0 * psi;
}

/* computeGreenStrainAndPiolaStressInversion  PositionBasedDynamics.cpp:1034-1104 */
static void PO(green_piola_inv)(const V3 *x, const M3 *im, R V0, R mu, R lambda, M3 *sigma, R *energy)
{
	M3 F = PO(defgrad)(x, im), U, VT, sD, eD, t, eps; R hF[3], eH[3], sv[3], trace, ltrace, psi = 0; int i, j;
	PO(svd_inv)(&F, hF, &U, &VT);
	for (j = 0; j < 3; j++) if (hF[j] < (R)0.577) hF[j] = (R)0.577;
	for (j = 0; j < 3; j++) eH[j] = (R)0.5 * (hF[j] * hF[j] - (R)1);
	trace = eH[0] + eH[1] + eH[2];
	ltrace = lambda * trace;
	for (j = 0; j < 3; j++) { sv[j] = eH[j] * (R)2.0 * mu; sv[j] += ltrace; sv[j] = hF[j] * sv[j]; }
	for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) { sD.m[i][j] = (i == j) ? sv[i] : 0; eD.m[i][j] = (i == j) ? eH[i] : 0; }
	t = PO(mmul)(&U, &eD); eps = PO(mmul)(&t, &VT);
	t = PO(mmul)(&U, &sD); *sigma = PO(mmul)(&t, &VT);
	for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) psi += eps.m[i][j] * eps.m[i][j];
	psi = mu * psi + (R)0.5 * lambda * trace * trace;
	*energy = V0 * psi;
}

/* computeGradCGreen  PositionBasedDynamics.cpp:1011-1031 */
static void PO(grad_c)(R V0, const M3 *im, const M3 *sigma, V3 *J)
{
	M3 T = PO(mtr)(im), H = PO(mmul)(sigma, &T); int i, j;
	for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) H.m[i][j] = H.m[i][j] * V0;
	J[0] = PO(col)(&H, 0); J[1] = PO(col)(&H, 1); J[2] = PO(col)(&H, 2);
	J[3] = SUB(SUB(NEG(J[0]), J[1]), J[2]);
}

/* FEMTetConstraint / XPBD_FEMTetConstraint: wrapper Constraints.cpp:1776-1825, 1851-1906;
 * solvers PositionBasedDynamics.cpp:1109-1169, XPBD.cpp:217-294; par = [V0, im(9), E, nu] */
static int PO(solve_fem_tet)(const V3 *p, const R *w, const R *par, int xpbd, R dt, R *mult, V3 *c)
{
	R V0 = par[0], E = par[10], nu = par[11], volume, sum, C = 0;
	M3 im = PO(load_cm)(&par[1]), sigma; V3 g[4]; int hi;
	{
		R cur = -(R)(1.0 / 6.0) * DOT(SUB(p[3], p[0]), CRS(SUB(p[2], p[0]), SUB(p[1], p[0])));
		hi = ((double)(cur / V0) < 0.2);
	}
	c[0] = MK(0, 0, 0); c[1] = c[0]; c[2] = c[0]; c[3] = c[0];
	if (E <= 0) return 1;
	if (nu < 0 || (double)nu > 0.49) return 0;
	volume = DOT(CRS(SUB(p[1], p[0]), SUB(p[2], p[0])), SUB(p[3], p[0])) / (R)6;
	if (!xpbd)
	{
		R mu = E / (R)2 / ((R)1 + nu);
		R lambda = E * nu / ((R)1 + nu) / ((R)1 - (R)2 * nu), sc;
		if (!hi || volume > 0) PO(green_piola)(p, &im, V0, mu, lambda, &sigma, &C);
		else PO(green_piola_inv)(p, &im, V0, mu, lambda, &sigma, &C);
		PO(grad_c)(V0, &im, &sigma, g);
		sum = w[0] * SQN(g[0]) + w[1] * SQN(g[1]) + w[2] * SQN(g[2]) + w[3] * SQN(g[3]);
		if (sum < EPS) return 0;
		sc = C / sum;
		c[0] = SCL(-sc * w[0], g[0]); c[1] = SCL(-sc * w[1], g[1]); c[2] = SCL(-sc * w[2], g[2]); c[3] = SCL(-sc * w[3], g[3]);
		return 1;
	}
	{
		/* 1.0 is a double literal in the reference: these are double expressions narrowed to Real */
		R mu_ = (R)(1.0 / (double)(R)2 / (double)((R)1 + nu));
		R lambda_ = (R)(1.0 * (double)nu / (double)((R)1 + nu) / (double)((R)1 - (R)2 * nu));
		R U_ = 0, Cc, alpha, lam;
		if (!hi || volume > 0) PO(green_piola)(p, &im, V0, mu_, lambda_, &sigma, &U_);
		else PO(green_piola_inv)(p, &im, V0, mu_, lambda_, &sigma, &U_);
		PO(grad_c)(V0, &im, &sigma, g);
		Cc = (R)sqrt(2.0 * (double)U_);
		sum = w[0] * SQN(g[0]) + w[1] * SQN(g[1]) + w[2] * SQN(g[2]) + w[3] * SQN(g[3]);
		alpha = (R)1 / (E * dt * dt);
		sum += Cc * Cc * alpha;
		if (sum < EPS) return 0;
		lam = -Cc * (Cc + alpha * *mult) / sum;
		*mult += lam;
		c[0] = SCL(lam * w[0], g[0]); c[1] = SCL(lam * w[1], g[1]); c[2] = SCL(lam * w[2], g[2]); c[3] = SCL(lam * w[3], g[3]);
	}
	return 1;
}

/* PositionBasedDynamics.cpp:713-805; par = [im(9), stretch, shear, normStretch, normShear] */
static int PO(solve_strain_tet)(const V3 *p, const R *w, const R *par, V3 *corr)
{
	M3 im = PO(load_cm)(&par[0]); V3 c[3]; int i, j, k;
	int nStretch = par[11] != 0, nShear = par[12] != 0;
	corr[0] = MK(0, 0, 0); corr[1] = corr[0]; corr[2] = corr[0]; corr[3] = corr[0];
	c[0] = PO(col)(&im, 0); c[1] = PO(col)(&im, 1); c[2] = PO(col)(&im, 2);
	for (i = 0; i < 3; i++) for (j = 0; j <= i; j++)
	{
		M3 P; V3 fi, fj, d[4]; R Sij, wi = 0, wj = 0, s1 = 0, s3 = 0, lambda; int ns = nShear && i != j;
		PO(setcol)(&P, 0, SUB(ADD(p[1], corr[1]), ADD(p[0], corr[0])));
		PO(setcol)(&P, 1, SUB(ADD(p[2], corr[2]), ADD(p[0], corr[0])));
		PO(setcol)(&P, 2, SUB(ADD(p[3], corr[3]), ADD(p[0], corr[0])));
		fi = PO(mvec)(&P, c[i]); fj = PO(mvec)(&P, c[j]);
		Sij = DOT(fi, fj);
		if (ns) { wi = NRM(fi); wj = NRM(fj); s1 = (R)1 / (wi * wj); s3 = s1 * s1 * s1; }
		d[0] = MK(0, 0, 0);
		for (k = 0; k < 3; k++)
		{
			d[k + 1] = ADD(SCR(fj, im.m[k][i]), SCR(fi, im.m[k][j]));
			if (ns)
				d[k + 1] = SUB(SCL(s1, d[k + 1]), SCL(Sij * s3, ADD(SCR(SCL(wj * wj, fi), im.m[k][i]), SCR(SCL(wi * wi, fj), im.m[k][j]))));
			d[0] = SUB(d[0], d[k + 1]);
		}
		if (ns) Sij *= s1;
		lambda = w[0] * SQN(d[0]) + w[1] * SQN(d[1]) + w[2] * SQN(d[2]) + w[3] * SQN(d[3]);
		if (PO_FABS(lambda) < EPS) continue;
		if (i == j)
		{
			if (nStretch) { R sq = PO_SQRT(Sij); lambda = (R)2 * sq * (sq - (R)1) / lambda * par[9]; }
			else lambda = (Sij - (R)1) / lambda * par[9];
		}
		else
			lambda = Sij / lambda * par[10];
		corr[0] = SUB(corr[0], SCL(lambda * w[0], d[0])); corr[1] = SUB(corr[1], SCL(lambda * w[1], d[1]));
		corr[2] = SUB(corr[2], SCL(lambda * w[2], d[2])); corr[3] = SUB(corr[3], SCL(lambda * w[3], d[3]));
	}
	return 1;
}

/* MathFunctions.cpp:147-175 */
static R PO(one_norm)(const M3 *A)
{
	/* ::fabs(double): each column sum is accumulated in double and narrowed to Real (MathFunctions.cpp:149-151) */
	R s1 = (R)(fabs((double)A->m[0][0]) + fabs((double)A->m[1][0]) + fabs((double)A->m[2][0]));
	R s2 = (R)(fabs((double)A->m[0][1]) + fabs((double)A->m[1][1]) + fabs((double)A->m[2][1]));
	R s3 = (R)(fabs((double)A->m[0][2]) + fabs((double)A->m[1][2]) + fabs((double)A->m[2][2]));
	R mx = s1; if (s2 > mx) mx = s2; if (s3 > mx) mx = s3; return mx;
}
static R PO(inf_norm)(const M3 *A)
{
	R s1 = (R)(fabs((double)A->m[0][0]) + fabs((double)A->m[0][1]) + fabs((double)A->m[0][2]));
	R s2 = (R)(fabs((double)A->m[1][0]) + fabs((double)A->m[1][1]) + fabs((double)A->m[1][2]));
	R s3 = (R)(fabs((double)A->m[2][0]) + fabs((double)A->m[2][1]) + fabs((double)A->m[2][2]));
	R mx = s1; if (s2 > mx) mx = s2; if (s3 > mx) mx = s3; return mx;
}

/* MathFunctions::polarDecompositionStable  MathFunctions.cpp:181-254 */
static void PO(polar_stable)(const M3 *M, R tol, M3 *Rm)
{
	M3 Mt = PO(mtr)(M), Adj, Et; R Mone = PO(one_norm)(M), Minf = PO(inf_norm)(M), Eone; int i, j;
	do
	{
		R det, Aone, Ainf, gamma, g1, g2;
		PO(setrow)(&Adj, 0, CRS(PO(row)(&Mt, 1), PO(row)(&Mt, 2)));
		PO(setrow)(&Adj, 1, CRS(PO(row)(&Mt, 2), PO(row)(&Mt, 0)));
		PO(setrow)(&Adj, 2, CRS(PO(row)(&Mt, 0), PO(row)(&Mt, 1)));
		det = Mt.m[0][0] * Adj.m[0][0] + Mt.m[0][1] * Adj.m[0][1] + Mt.m[0][2] * Adj.m[0][2];
		if ((double)PO_FABS(det) < 1.0e-12)
		{
			int index = -1; M3 M2;
			for (i = 0; i < 3; i++) { R len = SQN(PO(row)(&Adj, i)); if ((double)len > 1.0e-12) { index = i; break; } }
			if (index < 0) { for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) Rm->m[i][j] = (i == j); return; }
			PO(setrow)(&Mt, index, CRS(PO(row)(&Mt, (index + 1) % 3), PO(row)(&Mt, (index + 2) % 3)));
			PO(setrow)(&Adj, (index + 1) % 3, CRS(PO(row)(&Mt, (index + 2) % 3), PO(row)(&Mt, index)));
			PO(setrow)(&Adj, (index + 2) % 3, CRS(PO(row)(&Mt, index), PO(row)(&Mt, (index + 1) % 3)));
			M2 = PO(mtr)(&Mt);
			Mone = PO(one_norm)(&M2); Minf = PO(inf_norm)(&M2);
			det = Mt.m[0][0] * Adj.m[0][0] + Mt.m[0][1] * Adj.m[0][1] + Mt.m[0][2] * Adj.m[0][2];
		}
		Aone = PO(one_norm)(&Adj); Ainf = PO(inf_norm)(&Adj);
		gamma = (R)sqrt(sqrt((double)((Aone * Ainf) / (Mone * Minf))) / fabs((double)det));   /* double ::sqrt / ::fabs, MathFunctions.cpp:235 */
		g1 = gamma * (R)0.5;
		g2 = (R)0.5 / (gamma * det);
		for (i = 0; i < 3; i++) for (j = 0; j < 3; j++)
		{
			Et.m[i][j] = Mt.m[i][j];
			Mt.m[i][j] = g1 * Mt.m[i][j] + g2 * Adj.m[i][j];
			Et.m[i][j] -= Mt.m[i][j];
		}
		Eone = PO(one_norm)(&Et);
		Mone = PO(one_norm)(&Mt); Minf = PO(inf_norm)(&Mt);
	} while (Eone > Mone * tol);
	*Rm = PO(mtr)(&Mt);
}

/* solve_ShapeMatchingConstraint  PositionBasedDynamics.cpp:501-558; par = [k, restCm(3), x0(12), w(4), nc(4)] */
static int PO(solve_shape_matching)(const V3 *x, const R *par, V3 *corr)
{
	V3 cm = MK(0, 0, 0), restCm = MK(par[1], par[2], par[3]), x0[4]; R wsum = 0; M3 mat, Rm; int i;
	for (i = 0; i < 4; i++) { corr[i] = MK(0, 0, 0); x0[i] = MK(par[4 + 3 * i], par[5 + 3 * i], par[6 + 3 * i]); }
	for (i = 0; i < 4; i++) { R wi = (R)1 / (par[16 + i] + EPS); cm = ADD(cm, SCR(x[i], wi)); wsum += wi; }
	if (wsum == 0) return 0;
	cm = DVS(cm, wsum);
	memset(&mat, 0, sizeof(mat));
	for (i = 0; i < 4; i++)
	{
		V3 q = SUB(x0[i], restCm), p = SUB(x[i], cm); R wi = (R)1 / (par[16 + i] + EPS);
		p = SCR(p, wi);
		mat.m[0][0] += p.x * q.x; mat.m[0][1] += p.x * q.y; mat.m[0][2] += p.x * q.z;
		mat.m[1][0] += p.y * q.x; mat.m[1][1] += p.y * q.y; mat.m[1][2] += p.y * q.z;
		mat.m[2][0] += p.z * q.x; mat.m[2][1] += p.z * q.y; mat.m[2][2] += p.z * q.z;
	}
	PO(polar_stable)(&mat, EPS, &Rm);
	for (i = 0; i < 4; i++)
	{
		V3 goal = ADD(cm, PO(mvec)(&Rm, SUB(x0[i], restCm)));
		corr[i] = SCR(SUB(goal, x[i]), par[0]);
	}
	return 1;
}

/* one Constraint::solvePositionConstraint (wrappers in Simulation/Constraints.cpp:1181-2028) */
static void PO(solve_one)(PO(sim) *s, PO(con) *c, unsigned iter)
{
	unsigned nb = PO(nbodies)[c->type], i;
	V3 p[4], corr[4]; R w[4]; int res = 0;
	const R dt = s->h;
	for (i = 0; i < nb; i++) { p[i] = PO(ld)(s->x, c->b[i]); w[i] = s->w[c->b[i]]; }
	switch (c->type)
	{
	case 0: res = PO(solve_distance)(p[0], w[0], p[1], w[1], c->p[0], c->p[1], &corr[0], &corr[1]); break;
	case 1: if (iter == 0) c->lambda = 0;
		res = PO(solve_distance_xpbd)(p[0], w[0], p[1], w[1], c->p[0], c->p[1], dt, &c->lambda, &corr[0], &corr[1]); break;
	case 2: res = PO(solve_dihedral)(p, w, c->p[0], c->p[1], corr); break;
	case 3: res = PO(solve_isometric)(p, w, &c->p[1], c->p[0], 0, dt, NULL, corr); break;
	case 4: if (iter == 0) c->lambda = 0;
		res = PO(solve_isometric)(p, w, &c->p[1], c->p[0], 1, dt, &c->lambda, corr); break;
	case 5: res = PO(solve_fem_triangle)(p, w, c->p, corr); break;
	case 6: res = PO(solve_strain_triangle)(p, w, c->p, corr); break;
	case 7: res = PO(solve_volume)(p, w, c->p[0], c->p[1], 0, dt, NULL, corr); break;
	case 8: if (iter == 0) c->lambda = 0;
		res = PO(solve_volume)(p, w, c->p[0], c->p[1], 1, dt, &c->lambda, corr); break;
	case 9: res = PO(solve_fem_tet)(p, w, c->p, 0, dt, NULL, corr); break;
	case 10: if (iter == 0) c->lambda = 0;
		res = PO(solve_fem_tet)(p, w, c->p, 1, dt, &c->lambda, corr); break;
	case 11: res = PO(solve_strain_tet)(p, w, c->p, corr); break;
	case 12:
		res = PO(solve_shape_matching)(p, c->p, corr);
		if (res)
			for (i = 0; i < 4; i++)
				if (c->p[16 + i] != 0)   /* m_w captured at init, Constraints.cpp:2021-2025 */
				{
					R f = (R)(1.0 / (double)(unsigned)c->p[20 + i]);
					PO(st)(s->x, c->b[i], ADD(p[i], SCL(f, corr[i])));
				}
		return;
	default: break;
	}
	if (res)
		for (i = 0; i < nb; i++)
			if (w[i] != 0) PO(st)(s->x, c->b[i], ADD(p[i], corr[i]));
}

void PO(solve_position_constraints)(PO(sim) *s, unsigned iter)
{
	unsigned i;
	for (i = 0; i < s->nc; i++) PO(solve_one)(s, &s->c[i], iter);
}

/* TimeStepController::step for particle scenes  TimeStepController.cpp:75-241 */
void PO(step)(PO(sim) *s, unsigned nsteps)
{
	unsigned st, sub, it, g, i, k;
	PO(init_constraint_groups)(s);
	for (st = 0; st < nsteps; st++)
	{
		const R hOld = s->h;
		R h;
		/* TimeStep::clearAccelerations  TimeStep.cpp:28-62 */
		for (i = 0; i < s->n; i++) if (s->mass[i] != 0) { s->a[3 * i] = s->g[0]; s->a[3 * i + 1] = s->g[1]; s->a[3 * i + 2] = s->g[2]; }
		h = hOld / (R)s->sub_steps;
		s->h = h;     /* XPBD constraints read dt from the TimeManager, TimeStepController.cpp:92 */
		for (sub = 0; sub < s->sub_steps; sub++)
		{
			for (i = 0; i < s->n; i++)   /* :112-118 + TimeIntegration.cpp:7-19 */
			{
				for (k = 0; k < 3; k++) { s->lx[3 * i + k] = s->ox[3 * i + k]; s->ox[3 * i + k] = s->x[3 * i + k]; }
				if (s->mass[i] != 0)
					for (k = 0; k < 3; k++)
					{
						s->v[3 * i + k] = s->v[3 * i + k] + s->a[3 * i + k] * h;
						s->x[3 * i + k] = s->x[3 * i + k] + s->v[3 * i + k] * h;
					}
			}
			for (it = 0; it < s->max_iter; it++)   /* positionConstraintProjection :251-295 */
				for (g = 0; g < s->ng; g++)
					for (i = 0; i < s->gsz[g]; i++) PO(solve_one)(s, &s->c[s->grp[g][i]], it);
			{
				const R ih = (R)(1.0 / (double)h);   /* TimeIntegration.cpp:50,78: (1.0 / h) is a double expression */
				for (i = 0; i < s->n; i++)
				{
					if (s->mass[i] == 0) continue;
					for (k = 0; k < 3; k++)
					{
						if (s->vel_method == 0) s->v[3 * i + k] = ih * (s->x[3 * i + k] - s->ox[3 * i + k]);
						else s->v[3 * i + k] = ih * ((R)1.5 * s->x[3 * i + k] - (R)2.0 * s->ox[3 * i + k] + (R)0.5 * s->lx[3 * i + k]);
					}
				}
			}
		}
		s->h = hOld;
		s->time = s->time + hOld;
	}
}

/* ---- accessors ---------------------------------------------------------------------------- */
int PO(real_size)(void) { return (int)sizeof(R); }
unsigned PO(num_particles)(const PO(sim) *s) { return s->n; }
unsigned PO(num_constraints)(const PO(sim) *s) { return s->nc; }
int PO(constraint_type)(const PO(sim) *s, unsigned c) { return s->c[c].type; }
void PO(constraint_bodies)(const PO(sim) *s, unsigned c, unsigned *out) { memcpy(out, s->c[c].b, PO(nbodies)[s->c[c].type] * sizeof(unsigned)); }
int PO(constraint_params)(const PO(sim) *s, unsigned c, double *out)
{
	unsigned i, n = PO(nparams)[s->c[c].type];
	for (i = 0; i < n; i++) out[i] = (double)s->c[c].p[i];
	return (int)n;
}
double PO(constraint_lambda)(const PO(sim) *s, unsigned c) { return (double)s->c[c].lambda; }
unsigned PO(num_groups)(PO(sim) *s) { PO(init_constraint_groups)(s); return s->ng; }
unsigned PO(group_size)(const PO(sim) *s, unsigned g) { return s->gsz[g]; }
void PO(get_group)(const PO(sim) *s, unsigned g, unsigned *out) { memcpy(out, s->grp[g], s->gsz[g] * sizeof(unsigned)); }
void PO(set_params)(PO(sim) *s, unsigned sub, unsigned it, int vel) { s->sub_steps = sub < 1 ? 1 : sub; s->max_iter = it < 1 ? 1 : it; s->vel_method = vel; }
void PO(set_time_step_size)(PO(sim) *s, double h) { s->h = (R)h; }
void PO(set_gravity)(PO(sim) *s, double x, double y, double z) { s->g[0] = (R)x; s->g[1] = (R)y; s->g[2] = (R)z; }
double PO(get_time)(const PO(sim) *s) { return (double)s->time; }

static R *PO(arr)(PO(sim) *s, int which)
{
	switch (which) { case 0: return s->x; case 1: return s->x0; case 2: return s->v; case 3: return s->a; case 4: return s->ox; case 5: return s->lx; case 6: return s->mass; default: return s->w; }
}
void PO(get_array)(PO(sim) *s, int which, double *out)
{
	unsigned i, n = which < 6 ? 3 * s->n : s->n; const R *a = PO(arr)(s, which);
	for (i = 0; i < n; i++) out[i] = (double)a[i];
}
void PO(set_array)(PO(sim) *s, int which, const double *in)
{
	unsigned i;
	if (which == 6) { for (i = 0; i < s->n; i++) PO(set_mass)(s, i, in[i]); return; }
	if (which == 7) return;
	{ R *a = PO(arr)(s, which); for (i = 0; i < 3 * s->n; i++) a[i] = (R)in[i]; }
}
unsigned PO(tri_num_edges)(const PO(sim) *s, unsigned tm) { return s->tri[tm].ne; }
void PO(tri_get_edges)(const PO(sim) *s, unsigned tm, unsigned *out) { memcpy(out, s->tri[tm].edges, (size_t)s->tri[tm].ne * 4 * sizeof(unsigned)); }
unsigned PO(tet_num_edges)(const PO(sim) *s, unsigned tm) { return s->tet[tm].ne; }
void PO(tet_get_edges)(const PO(sim) *s, unsigned tm, unsigned *out) { memcpy(out, s->tet[tm].edges, (size_t)s->tet[tm].ne * 2 * sizeof(unsigned)); }

#undef V3
#undef M3
#undef MK
#undef ADD
#undef SUB
#undef NEG
#undef SCL
#undef SCR
#undef DVS
#undef DOT
#undef SQN
#undef NRM
#undef CRS
#undef NMZ
#undef EPS
#undef R
