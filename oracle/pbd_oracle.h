/* oracle/pbd_oracle.h -- TEST INFRASTRUCTURE ONLY.  C API of the plain-C oracle port; the same
 * set of functions exists with prefix po32_ (float) and po64_ (double).  All values cross the
 * API as double.  See pbd_oracle_impl.h for the reference citations. */
#ifndef PBD_ORACLE_H
#define PBD_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

#define PBD_ORACLE_DECLARE(P) \
	struct P##sim_s; \
	struct P##sim_s *P##create(void); \
	void P##destroy(struct P##sim_s *s); \
	int P##real_size(void); \
	unsigned P##add_vertex(struct P##sim_s *s, const double *p); \
	void P##set_mass(struct P##sim_s *s, unsigned i, double m); \
	int P##add_triangle_model(struct P##sim_s *s, unsigned np, unsigned nf, const double *pts, const unsigned *idx); \
	int P##add_regular_triangle_model(struct P##sim_s *s, int w, int h, const double *T, const double *R, const double *scale); \
	int P##add_tet_model(struct P##sim_s *s, unsigned np, unsigned nt, const double *pts, const unsigned *idx); \
	int P##add_regular_tet_model(struct P##sim_s *s, int w, int h, int d, const double *T, const double *R, const double *scale); \
	int P##add_constraint(struct P##sim_s *s, int type, const unsigned *bodies, const double *args, const unsigned *nclusters); \
	void P##add_cloth_constraints(struct P##sim_s *s, unsigned tm, unsigned method, double k, double xx, double yy, double xy, double xyP, double yxP, int ns, int nsh); \
	void P##add_bending_constraints(struct P##sim_s *s, unsigned tm, unsigned method, double k); \
	void P##add_solid_constraints(struct P##sim_s *s, unsigned tm, unsigned method, double k, double nu, double kv, int ns, int nsh); \
	void P##init_constraint_groups(struct P##sim_s *s); \
	void P##solve_position_constraints(struct P##sim_s *s, unsigned iter); \
	void P##step(struct P##sim_s *s, unsigned nsteps); \
	unsigned P##num_particles(const struct P##sim_s *s); \
	unsigned P##num_constraints(const struct P##sim_s *s); \
	int P##constraint_type(const struct P##sim_s *s, unsigned c); \
	void P##constraint_bodies(const struct P##sim_s *s, unsigned c, unsigned *out); \
	int P##constraint_params(const struct P##sim_s *s, unsigned c, double *out); \
	double P##constraint_lambda(const struct P##sim_s *s, unsigned c); \
	unsigned P##num_groups(struct P##sim_s *s); \
	unsigned P##group_size(const struct P##sim_s *s, unsigned g); \
	void P##get_group(const struct P##sim_s *s, unsigned g, unsigned *out); \
	void P##set_params(struct P##sim_s *s, unsigned sub, unsigned it, int vel); \
	void P##set_time_step_size(struct P##sim_s *s, double h); \
	void P##set_gravity(struct P##sim_s *s, double x, double y, double z); \
	double P##get_time(const struct P##sim_s *s); \
	void P##get_array(struct P##sim_s *s, int which, double *out); \
	void P##set_array(struct P##sim_s *s, int which, const double *in); \
	unsigned P##tri_num_edges(const struct P##sim_s *s, unsigned tm); \
	void P##tri_get_edges(const struct P##sim_s *s, unsigned tm, unsigned *out); \
	unsigned P##tet_num_edges(const struct P##sim_s *s, unsigned tm); \
	void P##tet_get_edges(const struct P##sim_s *s, unsigned tm, unsigned *out);

PBD_ORACLE_DECLARE(po32_)
PBD_ORACLE_DECLARE(po64_)

#ifdef __cplusplus
}
#endif
#endif
