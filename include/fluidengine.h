/*
 * fluidengine.h — C ABI of the MI355X-native FluidEngine MLS-MPM core.
 *
 * This is the drop-in boundary for the hot path of zhouxian/FluidLab:
 * everything `MPMSimulator` (fluidlab/fluidengine/simulators/mpm_simulator.py)
 * did through Taichi kernels is reached through these entry points.  The
 * reference has no FFI of its own (its "operators" are @ti.kernel methods
 * inlined by the Taichi JIT), so each entry point cites the reference method
 * it replaces.  Two libraries export exactly this symbol set:
 *
 *   fluidlab_amd/csrc/libfluidengine_hip.so   product: hand-written gfx950 HIP kernels (fe_real = float)
 *   oracle/_build/libfe_oracle_f32.so|_f64.so test infrastructure: CPU restatement of the reference
 *
 * Conventions
 *   - plain pointers and sizes only; host arrays are C-contiguous and borrowed
 *     for the duration of the call.  Particle arrays are always indexed by the
 *     caller's particle id (the engine may keep particles in another order).
 *   - every int-returning function returns 0 on success, non-zero on error;
 *     fe_last_error() gives the message.  Nothing aborts.
 *   - one engine <-> one device <-> one HIP stream; an engine is not
 *     thread-safe; engines are independent of each other.
 *   - frames f are *local* substep indices in [0, max_substeps_local]
 *     (mpm_simulator.py:225-252); f_global counts substeps since set_state.
 */
#ifndef FLUIDENGINE_H
#define FLUIDENGINE_H

#ifdef __cplusplus
extern "C" {
#endif

#ifndef FE_REAL
#define FE_REAL float
#endif
typedef FE_REAL fe_real;

typedef struct FeEngine FeEngine;

/* material classes — fluidlab/configs/macros.py:37-41 */
enum {
    FE_MAT_LIQUID              = 200,
    FE_MAT_PLASTO_ELASTIC      = 201,
    FE_MAT_ELASTIC             = 202,
    FE_MAT_RIGID               = 203,
    FE_MAT_PLASTO_ELASTIC_DEMO = 204
};

/* boundaries — fluidlab/fluidengine/boundaries/boundaries.py:28-134 */
enum { FE_BOUNDARY_CUBE = 0, FE_BOUNDARY_CYLINDER = 1 };

typedef struct FeBoundary {
    int     type;            /* FE_BOUNDARY_*                                       */
    fe_real lower[3];        /* cube: lower corner; cylinder: (0, y_lower, 0)       */
    fe_real upper[3];        /* cube: upper corner; cylinder: (1, y_upper, 1)       */
    fe_real xz_center[2];    /* cylinder only                                       */
    fe_real xz_radius;       /* cylinder only                                       */
    fe_real restitution;     /* Boundary.__init__, boundaries.py:10-12              */
    int     lock_dims;       /* bit d set => v[d] = 0 (lock_dims list)              */
} FeBoundary;

/* MPMSimulator.__init__ / build — mpm_simulator.py:14-71 */
typedef struct FeConfig {
    int        struct_size;          /* sizeof(FeConfig), ABI check                 */
    int        n_grid;               /* 64 * quality, mpm:21                        */
    int        n_particles;          /* mpm:54                                      */
    int        max_substeps_local;   /* L: frames 0..L are addressable, mpm:27      */
    int        n_substeps;           /* substeps per step, mpm:30                   */
    int        max_action_steps;     /* horizon (action_buffer length), effector.py:48 */
    fe_real    dt;                   /* mpm:24                                      */
    fe_real    p_vol;                /* (dx/2)^2, mpm:25                            */
    fe_real    gravity[3];           /* mpm:19                                      */
    FeBoundary boundary;             /* mpm:39-45                                   */
    int        device;               /* HIP device ordinal (ignored by the oracle)  */
} FeConfig;

/* effectors — fluidlab/fluidengine/effectors/effector.py, injector.py */
enum { FE_EFF_PLAIN = 0, FE_EFF_INJECTOR = 1, FE_EFF_AIRCON = 2 };     /* Effector/Rigid, Injector, AirCon (aircon.py) */

typedef struct FeEffectorDesc {
    int        struct_size;
    int        type;                 /* FE_EFF_*                                    */
    int        action_dim;           /* 0, 3 or 6 (8 for FE_EFF_AIRCON); effector.py:43 */
    fe_real    action_scale_v[8];    /* effector.py:54                              */
    fe_real    action_scale_p[8];    /* effector.py:55                              */
    FeBoundary boundary;             /* the effector's own boundary, effector.py:62 */
    /* Injector only (injector.py:13-36) */
    int        flux;                 /* particles injected per substep              */
    fe_real    radius;
    fe_real    inject_v[3];          /* Injector: particle velocity; AirCon: blowing direction (aircon.py:14-26) */
    fe_real    inject_p[3];
    int        locally_random;       /* random_vector indexed by f (1) or f_global (0) */
    int        randomize_inject_v;
    int        random_length;        /* rows of random_vector                       */
} FeEffectorDesc;

/* ---- lifecycle -------------------------------------------------------- */
FeEngine*   fe_create(const FeConfig* cfg);            /* NULL on failure; fe_last_error(NULL) */
void        fe_destroy(FeEngine* h);
const char* fe_last_error(FeEngine* h);                /* h may be NULL (creation errors)      */
const char* fe_backend(void);                          /* "hip-gfx950", "oracle-f32", "oracle-f64" */
int         fe_real_size(void);                        /* sizeof(fe_real)                      */
int         fe_sync(FeEngine* h);                      /* wait for the engine's stream         */
int         fe_set_option(FeEngine* h, const char* name, double value);  /* tunables, see DESIGN.md */
/* The value an option has right now, wherever it came from (default, fe_set_option, an FE_* environment variable read at fe_create):
 * bench.py echoes the effective options into its line.  Returns 1 for a name the library does not know. */
int         fe_get_option(FeEngine* h, const char* name, double* value);

/* ---- particles: init_particles_kernel, mpm:136-175 -------------------- */
/* x[N,3]; used/mat/mat_cls/body_id[N] i32; mu/lam/rho[N].  mass = p_vol*rho.
 * Frame 0 gets v=0, C=0, F=I. */
int fe_init_particles(FeEngine* h, const fe_real* x, const int* used, const int* mat,
                      const int* mat_cls, const fe_real* mu, const fe_real* lam,
                      const fe_real* rho, const int* body_id);

/* ---- the hot path: substep / substep_grad, mpm:515-552 ---------------- */
/* act != 0 <=> `not is_none_action`: agent.act / agent.move run (mpm:318-324,507-513). */
int fe_substep(FeEngine* h, int f, int f_global, int act);
int fe_substep_grad(FeEngine* h, int f, int f_global, int act);
/* n consecutive substeps in one crossing: the loops at mpm:749-751 / mpm:761-763.
 * fe_step walks f0, f0+1, ...; fe_step_grad walks f0+n-1 down to f0.
 * Particle adjoints live in a ring of two frames (slot f & 1), not in L+1 of them: after fe_step_grad(f0, n) the adjoint of frame f0 is
 * defined and nothing else is -- the HIP engine hands the adjoints of the frames in between from one substep to the next in registers
 * (option fuse_bwd), so the slot of frame f0 + 1 is left incomplete, and fe_get_grad / fe_add_grad / fe_copy_grad / fe_loss_step_grad on a
 * frame whose slot is in that state FAIL rather than return it. */
int fe_step(FeEngine* h, int f0, int f_global0, int n, int act);
int fe_step_grad(FeEngine* h, int f0, int f_global0, int n, int act);
/* Batched environments (BASELINE's "batched envs"; the reference steps one env per process, fluidlab/optimizer/solver.py:23-59):
 * the same n substeps for n_env engines of this process in lockstep -- same frame indices, same `act`.  Engines on one device
 * whose scenes use the same kernel variants (no SDF colliders / mesh effectors / MAT_RIGID bodies) share ONE launch per phase
 * (gridDim.y = n_env); any other set of engines is stepped one after the other.  Results are those of the individual calls. */
int fe_step_batch(FeEngine** hs, int n_env, int f0, int f_global0, int n, int act);
int fe_step_grad_batch(FeEngine** hs, int n_env, int f0, int f_global0, int n, int act);

/* ---- state I/O: mpm:555-609, 646-719 ---------------------------------- */
/* NULL pointers are skipped.  x,v [N,3]; C,F [N,3,3]; used [N] i32. */
int fe_get_frame(FeEngine* h, int f, fe_real* x, fe_real* v, fe_real* C, fe_real* F, int* used);
int fe_set_frame(FeEngine* h, int f, const fe_real* x, const fe_real* v, const fe_real* C,
                 const fe_real* F, const int* used);
/* The same with DEVICE pointers (memory of the engine's own device): the state never leaves the GPU.  The reference's
 * readframe/setframe fill torch CUDA tensors in place for ckpt_dest='gpu' (mpm:805-829, 895-897); for the CPU oracle a
 * "device" pointer is a host pointer. */
int fe_get_frame_dev(FeEngine* h, int f, fe_real* x, fe_real* v, fe_real* C, fe_real* F, int* used);
int fe_set_frame_dev(FeEngine* h, int f, const fe_real* x, const fe_real* v, const fe_real* C,
                     const fe_real* F, const int* used);
int fe_copy_frame(FeEngine* h, int src, int dst);           /* mpm:588-595 */
int fe_copy_grad(FeEngine* h, int src, int dst);            /* mpm:597-604 */
int fe_reset_grad(FeEngine* h);                             /* mpm:203-205 (+ effectors, effector.py:76-82) */
int fe_reset_grad_till_frame(FeEngine* h, int f);           /* mpm:606-609 */
/* adjoint access used by losses and tests: particles.grad[f].{x,v,C,F} */
int fe_get_grad(FeEngine* h, int f, fe_real* gx, fe_real* gv, fe_real* gC, fe_real* gF);
int fe_add_grad(FeEngine* h, int f, const fe_real* gx, const fe_real* gv, const fe_real* gC,
                const fe_real* gF);
/* the same with DEVICE pointers (the reference's loss kernels write particles.grad on the device: loss.py, *_loss.py) */
int fe_add_grad_dev(FeEngine* h, int f, const fe_real* gx, const fe_real* gv, const fe_real* gC,
                    const fe_real* gF);
/* particles_i.mat, used by Recorder (recorder.py:59) */
int fe_get_mat(FeEngine* h, int* mat);

/* ---- effectors / agent: effector.py:157-283, injector.py:54-105 ------- */
/* returns the effector index (>=0) or -1.  random_vector is [random_length, flux, 3]
 * (injector.py:54-60) or NULL for FE_EFF_PLAIN. */
int fe_add_effector(FeEngine* h, const FeEffectorDesc* desc, const fe_real* random_vector);
int fe_eff_set_act_range(FeEngine* h, int e, const int* act_range, int n);        /* injector.py:62-68 */
/* state = pos[3], quat[4] (wxyz), act_id — effector.py:185-208, injector.py:190-213 */
int fe_eff_get_state(FeEngine* h, int e, int f, fe_real* state8);
int fe_eff_set_state(FeEngine* h, int e, int f, const fe_real* state8);
/* v[3], w[3] of frame f (checkpoint payload, effector.py:84-102) */
int fe_eff_get_vw(FeEngine* h, int e, int f, fe_real* v3, fe_real* w3);
int fe_eff_set_vw(FeEngine* h, int e, int f, const fe_real* v3, const fe_real* w3);
/* AirCon only: strength s[f] and radius r[f] (aircon.py:20-21,178-191); set by set_velocity from action[6], action[7]
 * times their action_scale (aircon.py:211-213) */
int fe_eff_get_sr(FeEngine* h, int e, int f, fe_real* s, fe_real* r);
int fe_eff_set_sr(FeEngine* h, int e, int f, fe_real s, fe_real r);
int fe_eff_set_action(FeEngine* h, int e, int s, int s_global, int n_substeps,
                      const fe_real* action);                                     /* effector.py:262-268 */
int fe_eff_set_action_grad(FeEngine* h, int e, int s, int s_global, int n_substeps); /* effector.py:270-274 */
int fe_eff_apply_action_p(FeEngine* h, int e, const fe_real* action_p);           /* effector.py:227-231 */
int fe_eff_apply_action_p_grad(FeEngine* h, int e);                               /* effector.py:233-234 */
/* grad is [(n+1), action_dim]: rows 0..n-1 = action_buffer.grad[s..s+n), row n = action_buffer_p.grad */
int fe_eff_get_action_grad(FeEngine* h, int e, int s, int n, fe_real* grad);      /* effector.py:276-283 */
int fe_agent_copy_frame(FeEngine* h, int src, int dst);                           /* agent.py:117-120 */
int fe_agent_copy_grad(FeEngine* h, int src, int dst);                            /* agent.py:122-125 */
int fe_agent_reset_grad_till_frame(FeEngine* h, int f);                            /* agent.py:127-130, effector.py:178-183 */
/* The collector of AgentPouring / AgentJetBot (collector_act_kernel, agent_pouring.py:30-41, agent_jetbot.py:33-43): in
 * every acting substep a used particle outside `b` (Boundary.is_out, boundaries.py:80-93 / 127-134) is marked unused in
 * frames f and f+1 and parked at NOWHERE in f+1.  mat < 0: every material (AgentPouring); else only particles of that
 * material id (AgentJetBot: WATER).  b == NULL removes the collector.  Where the agent's colliders act -- at the particles,
 * at the grid nodes (mpm:393-395) or both, Agent.collide_type (agent.py:17-26) -- is fe_set_option("collide_type", 1|2|3). */
int fe_agent_set_collector(FeEngine* h, const FeBoundary* b, int mat);

/* ---- static SDF colliders: fluidlab/fluidengine/meshes/static.py:25-104, statics.py, mesh.py:57-66,120-127 -------- */
/* A collider is a signed-distance voxel grid in its own frame plus the affine map world -> voxel coordinates
 * (Mesh.T_mesh_to_voxels after init_transform, mesh.py:121).  grid_op runs v = statics[i].collide(I*dx, v) for every
 * static in the order they were added (mpm:386-390): inside the surface (sdf <= 0) the inward normal velocity is removed
 * and Coulomb friction applied to the tangential part (static.py:82-103).  The normal is the normalised central
 * difference (delta = 1e-2 voxels) of the trilinear SDF, mapped back by inverse(T[:3,:3]) (static.py:52-80). */
typedef struct FeSdfDesc {
    int     struct_size;
    int     res;                     /* voxels is [res,res,res], C order                       */
    fe_real T_mesh_to_voxels[16];    /* row-major 4x4, world/mesh position -> voxel coordinates */
    fe_real friction;                /* FRICTION[material], macros.py:131-141                   */
    fe_real softness;                /* Mesh.softness (dynamic colliders only)                  */
} FeSdfDesc;
int fe_add_static(FeEngine* h, const FeSdfDesc* desc, const fe_real* voxels);       /* returns the static's index or -1 */
/* Rigid.setup_mesh (rigid.py:19-24): give effector e a Dynamic mesh (dynamic.py:29-122).  agent.collide then runs at
 * particle level inside g2p (mpm:418-422, collide_type 'particle', agent.py:17): the SDF is sampled in the effector's
 * frame at f, the collider velocity comes from the pose change f -> f+1, `softness` blends the contact in, friction > 10
 * sticks.  Its adjoint reaches the material velocity, the particle position and the effector pose at f and f+1
 * (hence 6-dof action gradients through move_kernel's quaternion update, effector.py:157-161). */
int fe_eff_set_mesh(FeEngine* h, int e, const FeSdfDesc* desc, const fe_real* voxels);

/* ---- smoke field: fluidlab/fluidengine/simulators/smoke_field.py ------------------------------------------------------ */
/* An Eulerian smoke/temperature solver on its own res^3 grid, stepped once per *step* (mpm:745-747, 765-767): free-space
 * mask (a y-slab minus the static colliders, 191-201), RK3 back-trace advection of velocity and temperature + the AirCon
 * impulse (203-232, 315-360), divergence with solid-wall mirroring (234-258), `solver_iters` Jacobi sweeps (130-143),
 * pressure-gradient subtraction (273-288); step_grad is the hand-derived adjoint of all of it (113-128).  Frames s are
 * local step indices in [0, max_steps_local]. */
typedef struct FeSmokeConfig {
    int     struct_size;
    int     res;                 /* smoke_field.py:17 (128)                         */
    int     solver_iters;        /* :20                                              */
    int     q_dim;               /* :21; q[0] is the temperature                     */
    int     max_steps_local;     /* MPMSimulator.max_steps_local, :39                */
    fe_real dt;                  /* :19 (0.03)                                       */
    fe_real decay;               /* :22 (stored, unused by the reference's kernels)  */
    fe_real high_T, low_T;       /* :23-24                                           */
    int     lower_y, higher_y;   /* free slab lower_y < j < higher_y, :25-26 (60, 68) */
} FeSmokeConfig;
int fe_smoke_create(FeEngine* h, const FeSmokeConfig* cfg);          /* __init__, setup_fields, init_fields (:57-93) */
int fe_smoke_step(FeEngine* h, int s, int f);                         /* step(s, f), :95-111; needs an FE_EFF_AIRCON  */
int fe_smoke_step_grad(FeEngine* h, int s, int f);                    /* step_grad(s, f), :113-128                    */
/* frame I/O (get_state/set_state/ckpt).  NULL pointers are skipped.  v, v_tmp [res^3,3]; div, p [res^3]; q [res^3,q_dim] */
int fe_smoke_get_frame(FeEngine* h, int s, fe_real* v, fe_real* v_tmp, fe_real* div, fe_real* p, fe_real* q);
int fe_smoke_set_frame(FeEngine* h, int s, const fe_real* v, const fe_real* v_tmp, const fe_real* div, const fe_real* p, const fe_real* q);
int fe_smoke_get_grad(FeEngine* h, int s, fe_real* gv, fe_real* gq);
int fe_smoke_add_grad(FeEngine* h, int s, const fe_real* gv, const fe_real* gq);   /* losses seed q.grad / v.grad here */
int fe_smoke_copy_frame(FeEngine* h, int src, int dst);              /* :145-152 */
int fe_smoke_copy_grad(FeEngine* h, int src, int dst);               /* :154-161 */
int fe_smoke_reset_grad(FeEngine* h);                                /* :163-166 */
int fe_smoke_reset_grad_till_frame(FeEngine* h, int s);              /* :168-171 */

/* ---- loss: shapematching_loss.py:64-93 -------------------------------- */
int fe_loss_alloc(FeEngine* h, int max_loss_steps);
int fe_loss_set_target(FeEngine* h, int s, const fe_real* x);                     /* target['x'][s], [N,3] */
int fe_loss_clear(FeEngine* h);                                                   /* loss.py:55-61 */
/* chamfer_loss[s] += sum_p [used[f,p] && mat[p]==matching_mat] |x[f,p]-tgt_s[p]|^2   (matching_mat < 0: every material,
 * latteartstir_loss.py:62-68);
 * step_loss[s] += chamfer_loss[s] * weight   (shapematching_loss.py:80-88) */
int fe_loss_step(FeEngine* h, int s, int f, int matching_mat, fe_real weight);
/* x.grad[f,p] += 2 (x-tgt) * weight * step_loss_grad  (adjoint of the two kernels above) */
int fe_loss_step_grad(FeEngine* h, int s, int f, int matching_mat, fe_real weight,
                      fe_real step_loss_grad);
int fe_loss_get(FeEngine* h, fe_real* step_loss, int n);                          /* step_loss[0..n) */

/* ---- mesh -> signed distance (utils/mesh.py:63-96; the third-party mesh_to_sdf 0.0.x call) --------------------------
 * Stateless (no engine handle).  sdf[i] = signed distance from points[i] to the triangle mesh (verts[nv,3] f32,
 * faces[nf,3] i32): the exact point-triangle distance, negative where the generalized winding number of the mesh around
 * the point exceeds 1/2 in magnitude (either face orientation).  The reference gets an approximation of the same function from mesh_to_sdf's virtual scans
 * (scan_count 100*res/64, scan_resolution 400, sign from the 11 nearest scan normals); compute_sdf_data samples it on the
 * res^3 lattice over [-0.6, 0.6]^3, voxelize_mesh / Voxels.is_filled on particle positions (bodies.py:187-210).
 * `device`: GPU ordinal for the HIP library, ignored by the oracle.  Always float, whatever fe_real is. */
int fe_mesh_sdf(int device, const float* verts, int nv, const int* faces, int nf,
                const float* points, long long n_points, float* sdf);

/* ---- measurement ------------------------------------------------------ */
typedef struct FeStats {
    long long n_used;          /* used particles in the last processed frame            */
    long long n_cells_touched; /* Nc: grid nodes inside some used particle's stencil    */
    long long n_blocks_active; /* 4^3-cell blocks holding those nodes                   */
    long long n_slow_path;     /* particles that missed their LDS tile (global atomics) */
    long long bytes_state;     /* device bytes held by the engine                       */
} FeStats;
int fe_get_stats(FeEngine* h, int f, FeStats* out);
/* The work list of frame f's particle order (diagnostics; zeros on an engine without work lists):
 * out[0] work items, [1] first item of the global-atomics tail, [2] active blocks, [3] workgroups of blocks with several items,
 * [4] single-item blocks, [5..11] items holding 1, 2-4, 5-8, 9-16, 17-32, 33-64, 65-128 particles, [12] occupied blocks,
 * [13] particles of blocks without a work item (option loose_max), [14] single-item blocks small enough for a quad unit (option quad_max),
 * [15] quad units in the order's scatter list, [16] / [17] work units (workgroups with something to do) of the scatter / gather unit list, [18] the scatter list is
 * packed (no idle halves: options pack_units, quad_fit), [19] / [20] big / small leftover items of odd item counts, [21] / [22] waves of at
 * most 7 / 8..21 particles, whose particles take nine / three lanes each (option lane_split; 0 when it is off), [23] reserved.
 * fe_get_work_stats_n writes min(n, FE_WORK_STATS) entries: the list has grown from round to round, and a caller built against an
 * earlier header passes a shorter buffer (fe_get_work_stats = the full FE_WORK_STATS entries). */
#define FE_WORK_STATS 24
int fe_get_work_stats_n(FeEngine* h, int f, long long* out, int n);
int fe_get_work_stats(FeEngine* h, int f, long long out[FE_WORK_STATS]);
/* HIP-event stopwatch on the engine's stream */
int    fe_timer_start(FeEngine* h);
double fe_timer_stop_ms(FeEngine* h);               /* records, waits, returns elapsed ms (<0 on error) */
/* per-kernel event timing: enable, run some substeps, then read back.
 * names is a '\n'-separated list written into buf; ms_total/launches have `cap` slots. */
int fe_profile_enable(FeEngine* h, int on);
int fe_profile_read(FeEngine* h, char* buf, int buf_len, double* ms_total, long long* launches, int cap);

#ifdef __cplusplus
}
#endif
#endif /* FLUIDENGINE_H */
