/* A C program that knows nothing about GPUs: it builds the LargePyramid scene through the reference's PUBLIC API
 * (solver2d_amd/scenes/scenes.c: s2CreateWorld, s2CreateBody, s2CreatePolygonShape ...) and steps it with s2World_Step.  Linked against oracle/_ref/libs2ref.so -- the unmodified reference sources plus the binding of
 * INTEGRATION.md (oracle/ref_hook.c) --, one call, s2ref_use_amd(), moves the solver inside s2World_Step onto the
 * MI355X; s2ref_use_amd_world() moves the narrow phase and the refit too (the resident world chain), and
 * s2ref_world_device_pairs(1) the pair query.  Built and run by tools/dropin_demo.sh on a box that has /root/reference's headers (the build container) or the
 * prebuilt library (the GPU box: only this file's own declarations are needed, see below).
 *
 *   gcc -O2 tools/dropin_demo.c -o gpurun_out/dropin_demo -Loracle/_ref -ls2ref -Wl,-rpath,$PWD/oracle/_ref -lm
 *   gpurun_out/dropin_demo 200 60 solver2d_amd/libs2amd.so [scene solverId velIters posIters settleSteps]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

/* the slice of include/solver2d/{id,types,geometry,solver2d}.h this program uses, restated so that it compiles on the
 * GPU box, where the reference's headers do not exist (layouts: id.h:12-40, types.h:20-139, geometry.h:44-50) */
typedef struct { int16_t index; uint16_t revision; } s2WorldId;
typedef struct { int32_t index; int16_t world; uint16_t revision; } s2BodyId;
typedef struct { float x, y; } s2Vec2;

/* scenes.c (this repository, public API only) builds the sample scenes headless */
s2WorldId s2scene_create(const char* name, int solverType, int p0, int p1);
void s2World_Step(s2WorldId worldId, float timeStep, int32_t velIters, int32_t posIters, _Bool warmStart); /* solver2d.h:25 */
void s2DestroyWorld(s2WorldId id);
int s2ref_use_amd(const char* libraryPath, int device);
int s2ref_use_amd_world(const char* libraryPath, int device);
int s2ref_replace_error(void);
void s2ref_world_timing(double out[6]);
void s2ref_world_device_pairs(int on);
int s2ref_world_sizes(s2WorldId id, int32_t* bodies, int32_t* contacts, int32_t* joints);

static double now(void)
{
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static const char* g_scene = "pyramid";
static int g_solver = 7; /* s2_solverTGS_Soft, enum s2SolverType, types.h:75-88 */
static int g_vel = 8, g_pos = 4, g_settle = 45; /* long enough for the one-off search for a better strip partition (32 quiet steps after the last graph change) to fall outside the timed steps */

static double run(int base, int steps, const char* lib, int whole)
{
	s2WorldId w = s2scene_create(g_scene, g_solver, base, 0);
	s2ref_world_device_pairs(whole == 2);
	if (lib != NULL && (whole ? s2ref_use_amd_world(lib, 0) : s2ref_use_amd(lib, 0)) != 0)
	{
		fprintf(stderr, "could not load %s\n", lib);
		exit(1);
	}
	for (int i = 0; i < g_settle; ++i)
	{
		s2World_Step(w, 1.0f / 60.0f, g_vel, g_pos, 1);
	}
	double phases[6];
	s2ref_world_timing(phases);
	double t0 = now();
	for (int i = 0; i < steps; ++i)
	{
		s2World_Step(w, 1.0f / 60.0f, g_vel, g_pos, 1);
	}
	double ms = 1e3 * (now() - t0) / steps;
	int32_t nb = 0, nc = 0, nj = 0;
	s2ref_world_sizes(w, &nb, &nc, &nj);
	printf("%-34s %s %d: %d body slots, %d contact slots, %.3f ms per s2World_Step%s\n", lib ? (whole == 2 ? "+ stage 1 pair query on the MI355X" : whole ? "stages 3, solve, 4 on the MI355X" : "solver on the MI355X") : "reference (1 thread)", g_scene, base,
		   nb, nc, ms, lib && s2ref_replace_error() ? "  (solver reported an error)" : "");
	s2ref_world_timing(phases);
	if (whole && phases[5] > 0)
	{
		printf("    per step: stages 1+2 on the host %.3f ms, new contacts to the device %.3f, s2amd_world_step %.3f, download %.3f, pools and trees %.3f\n",
			   phases[0] / phases[5], phases[1] / phases[5], phases[2] / phases[5], phases[3] / phases[5], phases[4] / phases[5]);
	}
	if (lib != NULL)
	{
		s2ref_use_amd_world(NULL, 0);
	}
	s2DestroyWorld(w);
	return ms;
}

int main(int argc, char** argv)
{
	int base = argc > 1 ? atoi(argv[1]) : 100;
	int steps = argc > 2 ? atoi(argv[2]) : 30;
	const char* lib = argc > 3 ? argv[3] : "solver2d_amd/libs2amd.so";
	/* optional: scene, solver id, velocity iterations, position iterations, settling steps */
	g_scene = argc > 4 ? argv[4] : g_scene;
	g_solver = argc > 5 ? atoi(argv[5]) : g_solver;
	g_vel = argc > 6 ? atoi(argv[6]) : g_vel;
	g_pos = argc > 7 ? atoi(argv[7]) : g_pos;
	g_settle = argc > 8 ? atoi(argv[8]) : g_settle;
	double cpu = run(base, steps, NULL, 0);
	double gpu = run(base, steps, lib, 0);
	double whole = run(base, steps, lib, 1);
	double all = run(base, steps, lib, 2);
	printf("s2World_Step with the solver on the GPU (broad phase, narrow phase, refit on the host): x%.1f\n", cpu / gpu);
	printf("s2World_Step with narrow phase, solver and refit on the GPU (trees and contact pool on the host): x%.1f\n", cpu / whole);
	printf("s2World_Step with the pair query on the GPU as well (trees maintained, not queried): x%.1f\n", cpu / all);
	return 0;
}
