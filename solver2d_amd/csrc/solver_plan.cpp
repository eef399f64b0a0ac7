// The ten reference drivers (s2Solve_*) as lists of Ops, recorded once per parameter set.
#include "solver_internal.h"

StepConsts makeConsts(const s2amdStepParams* p)
{
	// src/world.c:170-202
	StepConsts sc;
	sc.dt = p->dt;
	sc.iterations = p->velIters;
	sc.extraIterations = p->posIters;
	sc.warmStart = p->warmStart != 0 ? 1 : 0;
	sc.inv_dt = p->dt > 0.0f ? 1.0f / p->dt : 0.0f;
	int type = p->solverType;
	if (type == s2amd_solverXPBD || type == s2amd_solverTGS_Soft || type == s2amd_solverTGS_Sticky || type == s2amd_solverTGS_NGS ||
		type == s2amd_solverSoftStep)
	{
		sc.h = sc.dt / sc.iterations;
		sc.inv_h = sc.inv_dt * sc.iterations;
	}
	else
	{
		sc.h = sc.dt;
		sc.inv_h = sc.inv_dt;
	}
	sc.gravityX = p->gravity[0];
	sc.gravityY = p->gravity[1];
	return sc;
}


namespace
{

// ------------------------------------------------------------------------------------------------
// plans: each builder records the stage sequence of one reference s2Solve_* function
// ------------------------------------------------------------------------------------------------
struct PlanBuilder
{
	StepPlan& p;
	const StepConsts& sc;

	void op(int code, int kind = 0, float h = 0.0f, float inv_h = 0.0f, bool useBias = false, int flag = 0)
	{
		Op o;
		o.code = code, o.kind = kind, o.useBias = useBias ? 1 : 0, o.flag = flag;
		o.h = h, o.inv_h = inv_h, o.f0 = 0.0f, o.f1 = 0.0f;
		p.ops.push_back(o);
	}
	void integrateVelocities() { op(OP_INTEGRATE_VEL); }
	void integratePositions(float h) { op(OP_INTEGRATE_POS, 0, h); }
	void finalizePositions(int dynamicOnly = 0) { op(OP_FINALIZE, 0, 0.0f, 0.0f, false, dynamicOnly); }
	void jointSweep(int kind, float h, float inv_h, bool useBias) { op(OP_JOINT_SWEEP, kind, h, inv_h, useBias); }
	void warmStartContacts(int kind) { op(OP_WARM, kind); }
	void solveSoft(int kind, float inv_h, bool useBias)
	{
		op(OP_SOLVE_SOFT, kind, 0.0f, inv_h, useBias);
		p.solveSweeps += 1;
	}
	void solveRigid(int kind, float inv_h)
	{
		op(OP_SOLVE_RIGID, kind, 0.0f, inv_h);
		p.solveSweeps += 1;
	}
	void solveNGS()
	{
		op(OP_SOLVE_NGS);
		p.solveSweeps += 1;
	}
	void solveSticky(float inv_h, bool useBias)
	{
		op(OP_SOLVE_STICKY, 0, 0.0f, inv_h, useBias);
		p.solveSweeps += 1;
	}
	void prepareContacts(int kind, float h, float hertz) { p.prepContacts = kind, p.prepH = h, p.prepHertz = hertz; }
	void prepareJoints(int kind, float h, float hertz, bool warm) { p.prepJoints = kind, p.jprepH = h, p.jprepHertz = hertz, p.jprepWarm = warm ? 1 : 0; }
	void storeImpulses(int kind, float scale = 0.0f) { p.storeKind = kind, p.storeScale = scale; }

	// s2Solve_TGS_Soft (solve_tgs_soft.c:138-280) / s2Solve_SoftStep (solve_soft_step.c:182-311)
	void solveTgsSoft(bool fixedAnchors)
	{
		float h = sc.h, inv_h = sc.inv_h;
		float contactHertz = S2_MINF(S2_CONTACT_HERTZ, 0.25f * inv_h);
		float jointHertz = fixedAnchors ? S2_MINF(S2_JOINT_HERTZ, 0.25f * inv_h) : S2_MINF(S2_JOINT_HERTZ, 0.125f * inv_h);
		p.unpackH = h;
		prepareContacts(PREP_SOFT, h, contactHertz);
		prepareJoints(JPREP_SOFT, h, jointHertz, true);
		for (int substep = 0; substep < sc.iterations; ++substep)
		{
			integrateVelocities();
			if (sc.warmStart)
			{
				jointSweep(JSOLVE_WARM, h, inv_h, false);
				warmStartContacts(fixedAnchors ? WARM_FIXED : WARM_CURRENT);
			}
			jointSweep(JSOLVE_SOFT, h, inv_h, true);
			solveSoft(fixedAnchors ? SOFT_FIXED : SOFT_TGS, inv_h, true);
			integratePositions(h);
			if (sc.extraIterations > 0)
			{
				jointSweep(JSOLVE_SOFT, h, inv_h, false);
				solveSoft(fixedAnchors ? SOFT_FIXED : SOFT_TGS, inv_h, false);
			}
		}
		finalizePositions();
		storeImpulses(STORE_PLAIN);
	}

	// s2Solve_Jacobi (solve_jacobi.c:134-292) / s2Solve_PGS_Soft (solve_pgs_soft.c:127-242)
	void solveJacobiOrPgsSoft(bool jacobi)
	{
		float h = sc.dt, inv_h = sc.inv_dt;
		float contactHertz = S2_MINF(S2_CONTACT_HERTZ, 0.333f * inv_h);
		float jointHertz = S2_MINF(S2_JOINT_HERTZ, 0.5f * inv_h);
		p.unpackH = h;
		integrateVelocities();
		prepareContacts(PREP_SOFT, h, contactHertz);
		if (sc.warmStart)
		{
			warmStartContacts(WARM_CURRENT);
		}
		// prepare reads only poses, warm start writes only velocities: "prepare all, then warm start
		// in order" is the reference's interleaved loop (solve_jacobi.c:193-206)
		prepareJoints(JPREP_SOFT, h, jointHertz, sc.warmStart != 0);
		if (sc.warmStart)
		{
			jointSweep(JSOLVE_WARM, h, inv_h, false);
		}
		for (int iter = 0; iter < sc.iterations; ++iter)
		{
			jointSweep(JSOLVE_SOFT, h, inv_h, true);
			solveSoft(jacobi ? SOFT_JACOBI : SOFT_PGS, inv_h, true);
			if (jacobi)
			{
				op(OP_JACOBI_APPLY);
			}
		}
		integratePositions(h);
		for (int iter = 0; iter < sc.extraIterations; ++iter)
		{
			jointSweep(JSOLVE_SOFT, h, inv_h, false);
			solveSoft(jacobi ? SOFT_JACOBI : SOFT_PGS, inv_h, false);
			if (jacobi)
			{
				op(OP_JACOBI_APPLY);
			}
		}
		finalizePositions();
		storeImpulses(STORE_PLAIN);
	}

	// s2Solve_PGS: solve_pgs.c:125-213
	void solvePgs()
	{
		float h = sc.dt, inv_h = sc.inv_dt;
		p.unpackH = h;
		integrateVelocities();
		prepareContacts(PREP_PGS, h, 0.0f);
		if (sc.warmStart)
		{
			warmStartContacts(WARM_CURRENT);
		}
		prepareJoints(JPREP_PLAIN, h, 0.0f, sc.warmStart != 0);
		if (sc.warmStart)
		{
			jointSweep(JSOLVE_WARM, h, inv_h, false);
		}
		for (int iter = 0; iter < sc.iterations; ++iter)
		{
			jointSweep(JSOLVE_BAUMGARTE, h, inv_h, true);
			solveRigid(RIGID_BAUMGARTE, inv_h);
		}
		integratePositions(h);
		finalizePositions();
		storeImpulses(STORE_PLAIN);
	}

	// s2Solve_PGS_NGS: solve_pgs_ngs.c:149-255.  The reference stores the impulses before the NGS
	// sweeps (:232); the NGS sweeps never touch an impulse, so storing after them is the same.
	void solvePgsNgs()
	{
		float h = sc.dt, inv_h = sc.inv_dt;
		p.unpackH = h;
		integrateVelocities();
		prepareContacts(PREP_PGS, h, 0.0f);
		if (sc.warmStart)
		{
			warmStartContacts(WARM_CURRENT);
		}
		prepareJoints(JPREP_PLAIN, h, 0.0f, sc.warmStart != 0);
		if (sc.warmStart)
		{
			jointSweep(JSOLVE_WARM, h, inv_h, false);
		}
		for (int iter = 0; iter < sc.iterations; ++iter)
		{
			jointSweep(JSOLVE_PLAIN, h, inv_h, false);
			solveRigid(RIGID_PGS, inv_h);
		}
		integratePositions(h);
		for (int iter = 0; iter < sc.extraIterations; ++iter)
		{
			jointSweep(JSOLVE_POSITION, h, inv_h, false);
			solveNGS();
		}
		finalizePositions();
		storeImpulses(STORE_PLAIN);
	}

	// s2Solve_PGS_NGS_Block: solve_pgs_ngs_block.c:892-963
	void solveBlock()
	{
		float h = sc.dt, inv_h = sc.inv_dt;
		p.unpackH = h;
		integrateVelocities();
		prepareContacts(PREP_BLOCK, h, 0.0f);
		warmStartContacts(WARM_BLOCK); // always applied: solve_pgs_ngs_block.c:279-319
		prepareJoints(JPREP_PLAIN, h, 0.0f, sc.warmStart != 0);
		if (sc.warmStart)
		{
			jointSweep(JSOLVE_WARM, h, inv_h, false);
		}
		for (int iter = 0; iter < sc.iterations; ++iter)
		{
			jointSweep(JSOLVE_PLAIN, h, inv_h, false);
			op(OP_BLOCK_VEL);
			p.solveSweeps += 1;
		}
		integratePositions(h);
		for (int iter = 0; iter < sc.extraIterations; ++iter)
		{
			op(OP_BLOCK_POS); // contacts before joints here (:945-957)
			p.solveSweeps += 1;
			jointSweep(JSOLVE_POSITION, h, inv_h, false);
		}
		finalizePositions();
		storeImpulses(STORE_BLOCK);
	}

	// s2Solve_TGS_NGS: solve_tgs_ngs.c:207-317
	void solveTgsNgs()
	{
		float h = sc.h, inv_h = sc.inv_h;
		p.unpackH = h;
		prepareContacts(PREP_TGS, h, 0.0f);
		prepareJoints(JPREP_PLAIN, h, 0.0f, sc.warmStart != 0);
		for (int substep = 0; substep < sc.iterations; ++substep)
		{
			integrateVelocities();
			if (sc.warmStart)
			{
				jointSweep(JSOLVE_WARM, h, inv_h, false);
				warmStartContacts(WARM_CURRENT);
			}
			jointSweep(JSOLVE_PLAIN, h, inv_h, false);
			solveRigid(RIGID_TGS, inv_h);
			integratePositions(h);
			jointSweep(JSOLVE_POSITION, h, inv_h, false);
			solveNGS();
		}
		finalizePositions();
		storeImpulses(STORE_PLAIN);
	}

	// s2Solve_TGS_Sticky: solve_tgs_sticky.c:313-417
	void solveTgsSticky()
	{
		float h = sc.h, inv_h = sc.inv_h;
		p.unpackH = h;
		prepareJoints(JPREP_PLAIN, h, 0.0f, false);
		prepareContacts(PREP_STICKY, h, 0.0f);
		for (int substep = 0; substep < sc.iterations; ++substep)
		{
			integrateVelocities();
			jointSweep(JSOLVE_BAUMGARTE, h, inv_h, true);
			solveSticky(inv_h, true);
			integratePositions(h);
		}
		finalizePositions();
		for (int iter = 0; iter < sc.extraIterations; ++iter)
		{
			jointSweep(JSOLVE_BAUMGARTE, h, inv_h, false);
			solveSticky(inv_h, false);
		}
		storeImpulses(STORE_PLAIN);
	}

	// s2Solve_XPBD: solve_xpbd.c:342-530
	void solveXpbd()
	{
		int substepCount = sc.iterations;
		if (substepCount == 0 || sc.dt == 0.0f)
		{
			p.earlyOut = true;
			return;
		}
		float h = sc.dt / substepCount;
		float inv_h = 1.0f / h;
		p.unpackH = h;
		p.usesDq0 = true;
		prepareContacts(PREP_XPBD, h, 0.0f);
		prepareJoints(JPREP_XPBD, h, 0.0f, false);
		for (int substep = 0; substep < substepCount; ++substep)
		{
			op(OP_XPBD_INTEGRATE, 0, h);
			jointSweep(JSOLVE_XPBD, h, inv_h, false);
			op(OP_XPBD_POS, 0, h);
			op(OP_XPBD_PROJECT, 0, 0.0f, inv_h);
			op(OP_XPBD_VEL, 0, h);
			p.solveSweeps += 2;
		}
		finalizePositions(1);
		storeImpulses(STORE_SCALED, inv_h);
	}
};

} // namespace

void buildPlan(s2amdSolver* s, const s2amdStepParams* params)
{
	if (s->plan.valid && memcmp(&s->plan.params, params, sizeof(*params)) == 0)
	{
		return;
	}
	StepPlan& p = s->plan;
	p = StepPlan();
	p.params = *params;
	p.sc = makeConsts(params);
	PlanBuilder b{p, p.sc};
	switch (params->solverType)
	{
		case s2amd_solverJacobi:
			b.solveJacobiOrPgsSoft(true);
			break;
		case s2amd_solverPGS:
			b.solvePgs();
			break;
		case s2amd_solverPGS_NGS:
			b.solvePgsNgs();
			break;
		case s2amd_solverPGS_NGS_Block:
			b.solveBlock();
			break;
		case s2amd_solverPGS_Soft:
			b.solveJacobiOrPgsSoft(false);
			break;
		case s2amd_solverSoftStep:
			b.solveTgsSoft(true);
			break;
		case s2amd_solverTGS_Sticky:
			b.solveTgsSticky();
			break;
		case s2amd_solverTGS_Soft:
			b.solveTgsSoft(false);
			break;
		case s2amd_solverTGS_NGS:
			b.solveTgsNgs();
			break;
		case s2amd_solverXPBD:
			b.solveXpbd();
			break;
	}
	// the two soft-coefficient triples of a PREP_SOFT plan, with prepareContactsKernel's operations
	p.sc.softDiet = p.prepContacts == PREP_SOFT ? 1 : 0;
	for (int i = 0; i < 2; ++i)
	{
		float contactHertz = i == 1 ? 2.0f * p.prepHertz : p.prepHertz;
		const float zeta = 10.0f;
		float h = p.prepH;
		float omega = 2.0f * S2_PI * contactHertz;
		float cc = h * omega * (2.0f * zeta + h * omega);
		float biasCoefficient = omega / (2.0f * zeta + h * omega);
		float impulseCoefficient = 1.0f / (1.0f + cc);
		float massCoefficient = cc * impulseCoefficient;
		p.sc.softCoef[i][0] = biasCoefficient;
		p.sc.softCoef[i][1] = massCoefficient;
		p.sc.softCoef[i][2] = impulseCoefficient;
	}
	p.valid = true;
	s->planGeneration += 1;
}

// Message passing applies when the global part is contact-only without a sequential tail (tables
// valid) and the plan consists of velocity-level contact sweeps only (poses change in body kernels).
bool messageEligible(const s2amdSolver* s, int solverType)
{
	if (!s->optMessage || !s->msgTablesValid)
	{
		return false;
	}
	return solverType == s2amd_solverTGS_Soft || solverType == s2amd_solverSoftStep || solverType == s2amd_solverPGS ||
		   solverType == s2amd_solverPGS_Soft || solverType == s2amd_solverTGS_Sticky;
}
