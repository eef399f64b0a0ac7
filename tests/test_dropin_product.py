"""The drop-in as a product artefact (shim/Makefile): libsolver2d_amd.so = an unmodified solver2d checkout + the binding +
the link-time call sites of shim/s2_amd_dropin.c.  CPU: it builds without anything from oracle/, exports the reference's
whole public API, and refuses loudly to step without the HIP library.  GPU: a C program written against the public headers
(tools/dropin_product_demo.c) steps a world through it, every route ending in the same bits where they must."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "shim", "_build", "libsolver2d_amd.so")
DEMO = os.path.join(ROOT, "tools", "dropin_product_demo.bin")
HAVE_REF = os.path.exists("/root/reference/src/world.c")

# include/solver2d/solver2d.h:22-70 ...
PUBLIC_API = """s2CreateWorld s2DestroyWorld s2World_Step s2World_Draw s2World_GetStatistics s2CreateBody s2DestroyBody s2Body_GetPosition
s2Body_GetAngle s2Body_GetLocalPoint s2Body_SetLinearVelocity s2Body_SetAngularVelocity s2Body_ApplyForceToCenter s2Body_ApplyLinearImpulse
s2Body_GetType s2Body_GetMass s2CreateCircleShape s2CreateSegmentShape s2CreateCapsuleShape s2CreatePolygonShape s2Shape_GetBody s2Shape_TestPoint
s2CreateMouseJoint s2CreateRevoluteJoint s2DestroyJoint s2MouseJoint_SetTarget s2RevoluteJoint_EnableLimit s2RevoluteJoint_EnableMotor
s2RevoluteJoint_SetMotorSpeed s2RevoluteJoint_GetMotorTorque s2World_QueryAABB""".split()
# ... and the non-inline helpers the samples use (geometry.h:82-108, hull.h:20, manifold.h:56-85, distance.h, dynamic_tree.h, math.h)
HELPERS = """s2MakeBox s2MakeSquare s2MakeOffsetBox s2MakeCapsule s2MakePolygon s2ComputeHull s2ComputeCircleMass s2ComputeCapsuleMass s2ComputePolygonMass
s2ComputeCircleAABB s2ComputeCapsuleAABB s2ComputePolygonAABB s2ComputeSegmentAABB s2PointInCircle s2PointInCapsule s2PointInPolygon
s2CollideCircles s2CollideCapsuleAndCircle s2CollidePolygonAndCircle s2CollidePolygons s2ShapeDistance s2SegmentDistance s2MakeProxy
s2DynamicTree_Create s2DynamicTree_Destroy s2DynamicTree_CreateProxy s2DynamicTree_DestroyProxy s2DynamicTree_MoveProxy s2DynamicTree_Query
s2DynamicTree_RayCast s2DynamicTree_Rebuild s2IsValid s2IsValidVec2 s2Normalize""".split()


@pytest.fixture(scope="module")
def built():
    if HAVE_REF:
        out = subprocess.run(["make", "-n", "-B", "-C", os.path.join(ROOT, "shim")], stdout=subprocess.PIPE, check=True).stdout.decode()
        assert "oracle" not in out and "tests/" not in out, "the product build must not need the test rig"
        subprocess.check_call([os.path.join(ROOT, "tools", "dropin_product_demo.sh"), "build"])
    if not os.path.exists(LIB):
        pytest.skip("shim/_build/libsolver2d_amd.so is built where the reference checkout is (make -C shim REF=...)")
    return LIB


def test_exports_the_whole_public_api(built):
    syms = set(re.findall(r" T (\w+)", subprocess.run(["nm", "-D", built], stdout=subprocess.PIPE, check=True).stdout.decode()))
    missing = [n for n in PUBLIC_API + HELPERS if n not in syms]
    assert not missing, missing
    # the link-time call sites are inside: s2World_Step is the drop-in's, the reference's lives on under its other name
    assert "s2World_Step_reference" in syms and "s2amdBinding_WorldStep" in syms
    needed = subprocess.run(["ldd", built], stdout=subprocess.PIPE, check=True).stdout.decode()
    assert "oracle" not in needed and "s2ref" not in needed


def test_the_patch_form_of_the_call_sites_applies_to_the_reference(tmp_path):
    if not HAVE_REF:
        pytest.skip("needs the reference checkout")
    for d in ("src", "include"):
        shutil.copytree(os.path.join("/root/reference", d), str(tmp_path / d))
    patch = open(os.path.join(ROOT, "shim", "call_sites.patch"), encoding="latin-1").read()
    assert not [l for l in patch.splitlines() if l.startswith("-") and not l.startswith("---")], "the patch only adds lines"
    # a device error must never fall through to the reference's CPU solver on the same world: the patched call sites abort
    assert "s2amdBinding_Solve(" not in patch and patch.count("s2amdBinding_SolveOrDie(") == 10 and "s2amdBinding_WorldStepOrDie(" in patch
    subprocess.run(["patch", "-p1", "-s"], input=patch.encode("latin-1"), cwd=str(tmp_path), check=True)
    for f in ("shim/s2_amd_binding.c", "shim/s2_amd_binding.h", "include/solver2d_amd.h"):
        shutil.copy(os.path.join(ROOT, f), str(tmp_path / "src"))
    srcs = sorted(str(p) for p in (tmp_path / "src").glob("*.c"))
    subprocess.check_call(["gcc", "-std=gnu17", "-O1", "-DNDEBUG", "-fPIC", "-w", "-I" + str(tmp_path / "include"), "-I" + str(tmp_path / "src"),
                           "-shared", "-o", str(tmp_path / "libpatched.so")] + srcs + ["-lm", "-ldl"])


def test_without_the_hip_library_the_first_step_fails_loudly(built):
    if not os.path.exists(DEMO):
        pytest.skip("demo binary not built")
    env = dict(os.environ, S2AMD_DROPIN="step", S2AMD_LIBRARY="/nonexistent/libs2amd.so")
    p = subprocess.run([DEMO, "10", "2", "pyramid", "7", "8", "4", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode != 0 and b"no CPU path" in p.stderr
    # ... and the reference's own solvers are one switch away
    p = subprocess.run([DEMO, "10", "2", "pyramid", "7", "8", "4", "1"], env=dict(os.environ, S2AMD_DROPIN="off"), stdout=subprocess.PIPE, check=True)
    assert b"state digest" in p.stdout


def test_every_pool_edit_of_the_public_api_reaches_the_binding(built):
    """residentMatches (shim/s2_amd_binding.c) compares pool counts: a destroy followed by a create leaves them equal, so every public
    function that creates or destroys a body or joint -- and every setter / reader of state a resident world keeps in HBM -- is the
    drop-in's own function (sync, then invalidate), with the reference's under <name>_reference.  A linker --wrap would not do: the
    program's calls come from outside the link."""
    syms = dict((name, kind) for kind, name in re.findall(r" ([TtU]) (\w+)", subprocess.run(["nm", "-D", built], stdout=subprocess.PIPE, check=True).stdout.decode()))
    for name in ("s2CreateBody", "s2DestroyBody", "s2DestroyJoint", "s2CreateMouseJoint", "s2CreateRevoluteJoint", "s2Body_SetLinearVelocity",
                 "s2Body_ApplyLinearImpulse", "s2MouseJoint_SetTarget", "s2World_QueryAABB", "s2World_Draw", "s2DestroyWorld", "s2CreatePolygonShape", "s2World_Step"):
        assert syms.get(name) == "T" and syms.get(name + "_reference") == "T", name
    # ... and the exported name really is the drop-in's code: it calls the binding
    dis = subprocess.run(["objdump", "-d", "--no-show-raw-insn", built], stdout=subprocess.PIPE, check=True).stdout.decode()
    for name in ("s2CreateBody", "s2DestroyJoint", "s2Body_SetLinearVelocity"):
        body = dis[dis.index("<%s>:" % name):]
        body = body[:body.index("\n\n")]
        assert "s2amdBinding_" in body or "editWorld" in body or "syncWorld" in body, name


def _digest(env_extra, args):
    env = dict(os.environ, S2AMD_LIBRARY=os.path.join(ROOT, "solver2d_amd", "libs2amd.so"), **env_extra)
    out = subprocess.run([DEMO] + [str(a) for a in args], env=env, stdout=subprocess.PIPE, check=True).stdout.decode()
    return re.search(r"state digest (\w+)", out).group(1), out


@pytest.mark.gpu
@pytest.mark.parametrize("scene,base,solver,vel,pos", [("pyramid", 40, 7, 8, 4), ("mixed", 24, 3, 4, 2), ("tumbler", 300, 0, 4, 2), ("joint_grid", 20, 2, 4, 2)])
def test_a_program_on_the_public_api_runs_on_the_gpu_through_the_product_library(built, scene, base, solver, vel, pos):
    if not os.path.exists(DEMO):
        pytest.skip("demo binary not built")
    args = [base, 30, scene, solver, vel, pos, 10]
    solver_only, _ = _digest({"S2AMD_DROPIN": "solver"}, args)
    step_host_pairs, _ = _digest({"S2AMD_DROPIN": "step", "S2AMD_DEVICE_PAIRS": "0"}, args)
    step_dev_pairs, out = _digest({"S2AMD_DROPIN": "step", "S2AMD_DEVICE_PAIRS": "1"}, args)
    # the whole-step routes are the solver-only route's computation, bit for bit (same contact slots, same sweep order)
    assert solver_only == step_host_pairs == step_dev_pairs, out


@pytest.mark.gpu
@pytest.mark.parametrize("scene,steps,settle", [("rush", 40, 20), ("warm_start_energy", 30, 100), ("ragdoll_stress", 160, 0)])
def test_samples_that_edit_the_world_every_frame_through_the_product_library(built, scene, steps, settle):
    """The three reference samples whose Step override touches the world (solver2d_amd/scenes/scenes.c: s2scene_pre_step / _post_step):
    Rush applies a force to each of its 400 bodies before every step, Warm Start Energy destroys a body at step 120, Ragdoll Stress
    creates a ragdoll (11 bodies, 14 shapes, 10 joints) every 30 steps.  Under s2Solve_Jacobi the contact pass does not depend on the
    sweep order, so the first two -- no joints -- must end in the reference's own bits on every route (option `incremental` 0: every
    body's sum in pool order, as the reference adds); the ragdolls' joints are swept colour by colour here and in pool order there, so
    for them the two whole-step routes are held against each other."""
    if not os.path.exists(DEMO):
        pytest.skip("demo binary not built")
    args = [0, steps, scene, 0, 4, 2, settle]
    opts = {"S2AMD_OPTIONS": "incremental=0"}
    host_pairs, out_h = _digest(dict(opts, S2AMD_DROPIN="step", S2AMD_DEVICE_PAIRS="0"), args)
    device_pairs, out_d = _digest(dict(opts, S2AMD_DROPIN="step", S2AMD_DEVICE_PAIRS="1"), args)
    assert host_pairs == device_pairs, (out_h, out_d)
    if scene != "ragdoll_stress":
        reference, _ = _digest({"S2AMD_DROPIN": "off"}, args)
        solver_only, out_s = _digest(dict(opts, S2AMD_DROPIN="solver"), args)
        assert reference == solver_only == device_pairs, (out_s, out_d)


def _edited(env_extra, args):
    digest, out = _digest(dict(env_extra, S2DEMO_EDITS="1"), args)
    m = re.search(r"edited body at \(([-\d.]+), ([-\d.]+)\) angle ([-\d.]+)", out)
    return digest, tuple(float(g) for g in m.groups()), out


@pytest.mark.gpu
@pytest.mark.parametrize("scene,base,solver,vel,pos", [("pyramid", 24, 7, 8, 4), ("mixed", 24, 3, 4, 2)])
def test_bodies_created_and_replaced_between_resident_steps(built, scene, base, solver, vel, pos):
    """A body created after lean steps lands in a slot the device holds as free (its record there is zeros); one destroyed and
    replaced leaves every pool count equal.  Every route must see both edits: the replacement body -- in free flight beside the scene -- ends
    exactly where the reference's own solver puts it (the rest of the world may be swept in another order after a re-upload, so the
    whole-world digests are not compared), and nothing of the zeroed device record may show (a body at the origin, a zero rotation)."""
    if not os.path.exists(DEMO):
        pytest.skip("demo binary not built")
    args = [base, 40, scene, solver, vel, pos, 10]
    _, ref, _ = _edited({"S2AMD_DROPIN": "off"}, args)
    plain, _ = _digest({"S2AMD_DROPIN": "solver"}, args)
    for env in ({"S2AMD_DROPIN": "solver"}, {"S2AMD_DROPIN": "step", "S2AMD_DEVICE_PAIRS": "0"}, {"S2AMD_DROPIN": "step", "S2AMD_DEVICE_PAIRS": "1"}):
        digest, got, out = _edited(env, args)
        assert digest != plain, "the edits change the world"
        # (in free flight beside the scene: the integrator's path, the same on every route -- the demo prints four decimals)
        assert got == ref and got[1] > 20.0, (env, got, ref, out)
