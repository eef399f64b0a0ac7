/* A C program written against the reference's PUBLIC headers only: it builds a sample scene (solver2d_amd/scenes/scenes.c,
 * public API) and calls s2World_Step.  Linked against shim/_build/libsolver2d_amd.so -- an unmodified solver2d checkout
 * + shim/s2_amd_binding.c + shim/s2_amd_dropin.c, built by shim/Makefile; nothing from oracle/ -- it runs on the MI355X
 * without a line of it knowing: the environment picks the route (S2AMD_DROPIN = off | solver | step, S2AMD_DEVICE_PAIRS).
 *
 *   tools/dropin_product_demo.sh [base steps scene solverId velIters posIters settleSteps]
 */
#include "solver2d/solver2d.h"
#include "solver2d/geometry.h"

#include <stdio.h>
#include <stdlib.h>
#include <time.h>

s2WorldId s2scene_create(const char* name, int solverType, int p0, int p1);
void s2scene_pre_step(s2WorldId w, int stepIndex, float timeStep);
void s2scene_post_step(s2WorldId w, float hertz);
void s2amdDropin_Timing(double out[6]);
unsigned long long s2amdDropin_StateDigest(s2WorldId worldId);

static double now(void)
{
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

int main(int argc, char** argv)
{
	const int base = argc > 1 ? atoi(argv[1]) : 100;
	const int steps = argc > 2 ? atoi(argv[2]) : 30;
	const char* scene = argc > 3 ? argv[3] : "pyramid";
	const int solver = argc > 4 ? atoi(argv[4]) : (int)s2_solverTGS_Soft;
	const int vel = argc > 5 ? atoi(argv[5]) : 8, pos = argc > 6 ? atoi(argv[6]) : 4;
	const int settle = argc > 7 ? atoi(argv[7]) : 45;
	const char* route = getenv("S2AMD_DROPIN") ? getenv("S2AMD_DROPIN") : "step";
	s2WorldId w = s2scene_create(scene, solver, base, 0);
	/* (a frame of a sample = what its Step override does before the step -- Warm Start Energy destroys a body, Rush applies forces --,
	 * the step, what it does after -- Ragdoll Stress creates and destroys ragdolls: solver2d_amd/scenes/scenes.c) */
	int frame = 0;
	for (int i = 0; i < settle; ++i, ++frame)
	{
		s2scene_pre_step(w, frame, 1.0f / 60.0f);
		s2World_Step(w, 1.0f / 60.0f, vel, pos, true);
		s2scene_post_step(w, 60.0f);
	}
	double phases[6];
	s2amdDropin_Timing(phases);
	/* S2DEMO_EDITS=1: the pools are edited between steps the way a game edits them -- a body created while the world is resident
	 * (a slot the device still holds as free), then destroyed and replaced by another one (every pool count as before: only the
	 * edit itself can tell the binding) -- every route must still end in the same bits */
	const int edits = getenv("S2DEMO_EDITS") != NULL && atoi(getenv("S2DEMO_EDITS")) != 0;
	s2BodyId extra = s2_nullBodyId;
	const double t0 = now();
	for (int i = 0; i < steps; ++i)
	{
		if (edits && (i == steps / 4 || i == steps / 2))
		{
			if (i == steps / 2)
			{
				s2DestroyBody(extra);
			}
			s2BodyDef bd = s2_defaultBodyDef;
			bd.type = s2_dynamicBody;
			/* (beside the scene, in free flight until the run ends: its path is the integrator's alone, the same bits on every route) */
			bd.position = (s2Vec2){0.5f * (float)base + (i == steps / 2 ? 20.0f : 23.0f), 30.0f};
			bd.linearVelocity = (s2Vec2){0.25f, -3.0f};
			bd.angularVelocity = 0.5f;
			extra = s2CreateBody(w, &bd);
			s2ShapeDef sd = s2_defaultShapeDef;
			s2Polygon box = s2MakeSquare(0.4f);
			s2CreatePolygonShape(extra, &sd, &box);
		}
		s2scene_pre_step(w, frame, 1.0f / 60.0f);
		s2World_Step(w, 1.0f / 60.0f, vel, pos, true);
		s2scene_post_step(w, 60.0f);
		frame += 1;
	}
	const double ms = 1e3 * (now() - t0) / steps;
	s2amdDropin_Timing(phases);
	printf("route %-6s pairs %s  %s %d, solver %d %d/%d: %.3f ms per s2World_Step over %d steps, state digest %016llx\n", route,
		   getenv("S2AMD_DEVICE_PAIRS") ? getenv("S2AMD_DEVICE_PAIRS") : "1", scene, base, solver, vel, pos, ms, steps, s2amdDropin_StateDigest(w));
	if (edits)
	{
		/* after pool edits the routes need not sweep in the same order any more (a re-uploaded world builds its structure afresh):
		 * what they must agree on is the physics */
		const s2Vec2 p = s2Body_GetPosition(extra);
		printf("    edited body at (%.4f, %.4f) angle %.4f\n", p.x, p.y, s2Body_GetAngle(extra));
	}
	if (phases[5] > 0)
	{
		printf("    per step: stages 1+2 on the host %.3f ms, new contacts to the device %.3f, s2amd_world_step %.3f, download %.3f, pools and trees %.3f\n",
			   phases[0] / phases[5], phases[1] / phases[5], phases[2] / phases[5], phases[3] / phases[5], phases[4] / phases[5]);
	}
	s2DestroyWorld(w);
	return 0;
}
