"""Model constants of the MuJoCo-backed bodies against the reference's only data for them: the XML files under
emei/envs/mujoco/assets (missing #4).  `oracle/gen_golden.py:gen_model_constants` parsed the files into
tests/golden/model_constants_golden.npz (numbers as written, <default> resolved, nothing derived).  Here MuJoCo's documented
compiler rules are applied to those numbers — capsule mass / inertia from the geom sizes at density 1000 (inertiafromgeom),
settotalmass, fromto -> centre / half-length / axis, degree -> radian, pair friction = max of the two geoms, the documented
defaults for what a file leaves out — and the result is compared with
  (i)  the oracle's tables   (oracle.xml_constants: exported from the structs oracle/*.c computes with), and
  (ii) the kernels' tables   (emei_model_constants: exported from the constexpr objects the HIP kernels are compiled from),
both in the layout include/emei_hip.h documents.  Any hand-typed constant that disagrees with the XML fails here, on the CPU.
"""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN

RHO = 1000.0                                    # MuJoCo default geom density
SOLREF_TC, SOLIMP = 0.02, (0.9, 0.95, 0.001)    # MuJoCo defaults: solref (.02 1), solimp (.9 .95 .001)
DEFAULT_FRICTION, DEFAULT_GRAVITY_Z = 1.0, -9.81


@pytest.fixture(scope="module")
def xml():
    return np.load(os.path.join(GOLDEN, "model_constants_golden.npz"))


def capsule_mass(r, half):
    return RHO * (np.pi * r * r * 2 * half + 4.0 / 3.0 * np.pi * r**3)


def capsule_inertia_perp(r, half):
    """about an axis through the centre, perpendicular to the capsule's axis: cylinder + two hemispherical caps"""
    h, mcyl, msph = 2 * half, RHO * np.pi * r * r * 2 * half, RHO * 4.0 / 3.0 * np.pi * r**3
    return mcyl * (3 * r * r + h * h) / 12 + msph * (2 * r * r / 5 + h * h / 4 + 3 * h * r / 8)


def _or(v, default):
    return default if np.isnan(v) else v


def kernel_constants(env_id):
    from emei_amd import _lib as L

    buf = (C.c_double * 256)()
    n = L.lib().emei_model_constants(env_id, C.cast(buf, C.c_void_p), 256)
    assert n > 0, L.lib().emei_last_error()
    return np.array(buf[:n])


def _capsule(x, t, g, world_pos):
    """world end-sphere centres (2 x [x, z]), radius, half-length of capsule geom g of model t"""
    r = x[f"{t}_geom_size"][g, 0]
    ft = x[f"{t}_geom_fromto"][g]
    base = np.zeros(2) if x[f"{t}_coordinate_global"] else world_pos[x[f"{t}_geom_body"][g]]
    if not np.isnan(ft[0]):
        a, b = base + ft[[0, 2]], base + ft[[3, 5]]
        return np.stack([a, b]), r, np.linalg.norm(b - a) / 2  # the size's second entry is ignored when fromto is given
    c = base + x[f"{t}_geom_pos"][g][[0, 2]]
    aa = x[f"{t}_geom_axisangle"][g]
    assert np.allclose(aa[:3], [0, 1, 0])  # rotation about +y of the default +z axis
    ang = aa[3] * (np.pi / 180 if x[f"{t}_angle_degree"] else 1.0)
    axis, half = np.array([np.sin(ang), np.cos(ang)]), x[f"{t}_geom_size"][g, 1]
    return np.stack([c - half * axis, c + half * axis]), r, half


def planar_expected(x, t):
    """the emei_model_constants vector of a planar tree (cheetah "ch", hopper "hp") derived from the XML numbers"""
    nb = len(x[f"{t}_body_names"])
    parent, glob = x[f"{t}_body_parent"], bool(x[f"{t}_coordinate_global"])
    deg = np.pi / 180 if x[f"{t}_angle_degree"] else 1.0
    assert np.isnan(x[f"{t}_body_quat"]).all()  # no body is rotated at qpos0
    world = np.zeros((nb, 2))
    for b in range(nb):
        p = x[f"{t}_body_pos"][b][[0, 2]]
        world[b] = p if glob or parent[b] < 0 else world[parent[b]] + p
    # body frame of the builds = the anchor of the body's hinge (world, qpos0)
    jb, hinge = x[f"{t}_joint_body"], x[f"{t}_joint_is_hinge"]
    anchor = np.zeros((nb, 2))
    for b in range(nb):
        (j,) = [k for k in range(len(jb)) if jb[k] == b and hinge[k]]
        jp = np.nan_to_num(x[f"{t}_joint_pos"][j][[0, 2]])
        anchor[b] = jp if glob else world[b] + jp
        assert abs(abs(x[f"{t}_joint_axis"][j][1]) - 1) < 1e-15  # hinges about +-y: a planar tree
    caps = [g for g in range(len(x[f"{t}_geom_names"])) if x[f"{t}_geom_is_capsule"][g] and x[f"{t}_geom_body"][g] >= 0]
    floor = [g for g in range(len(x[f"{t}_geom_names"])) if not x[f"{t}_geom_is_capsule"][g]]
    assert len(floor) == 1
    ends, rad, gm, gi, gc = [], [], [], [], []
    for g in caps:
        e, r, half = _capsule(x, t, g, world)
        ends.append(e), rad.append(r), gm.append(capsule_mass(r, half)), gi.append(capsule_inertia_perp(r, half)), gc.append(e.mean(axis=0))
    gbody = [int(x[f"{t}_geom_body"][g]) for g in caps]
    gm, gi, gc = np.array(gm), np.array(gi), np.array(gc)
    scale = 1.0 if np.isnan(x[f"{t}_settotalmass"]) else float(x[f"{t}_settotalmass"]) / gm.sum()
    out = [-_or(x[f"{t}_gravity"][2], DEFAULT_GRAVITY_Z)]
    for b in range(nb):
        sel = [k for k in range(len(caps)) if gbody[k] == b]
        mb = gm[sel].sum()
        com = (gm[sel, None] * gc[sel]).sum(axis=0) / mb
        inertia = sum(gi[k] + gm[k] * ((gc[k] - com) ** 2).sum() for k in sel)
        org = anchor[b] - (anchor[parent[b]] if parent[b] >= 0 else 0.0)
        out += [mb * scale, *(com - anchor[b]), inertia * scale, *org]
    ffl = _or(x[f"{t}_geom_friction"][floor[0], 0], DEFAULT_FRICTION)
    for k, g in enumerate(caps):
        e = ends[k] - anchor[gbody[k]]
        e = e[np.lexsort((e[:, 1], e[:, 0]))]  # the two ends as a set: a build may list them in either order
        out += [gbody[k], *e[0], *e[1], rad[k], max(_or(x[f"{t}_geom_friction"][g, 0], DEFAULT_FRICTION), ffl)]
    aj = x[f"{t}_act_joint"]
    for k, j in enumerate(aj):
        assert x[f"{t}_joint_limited"][j] and x[f"{t}_act_ctrllimited"][k]
        lo, hi = x[f"{t}_joint_range"][j] * deg
        out += [_or(x[f"{t}_joint_stiffness"][j], 0.0), _or(x[f"{t}_joint_damping"][j], 0.0), _or(x[f"{t}_joint_armature"][j], 0.0), lo, hi,
                x[f"{t}_act_gear"][k]]
    # the root joints carry no spring, damper, armature or limit (the builds assume it)
    for j in range(len(jb)):
        if j not in aj:
            assert not x[f"{t}_joint_limited"][j]
            assert all(_or(x[f"{t}_joint_{key}"][j], 0.0) == 0.0 for key in ("stiffness", "damping", "armature"))
    g0 = caps[0]
    for g in caps:  # one solref / solimp / margin for every capsule
        for key in ("solref", "solimp"):
            assert np.array_equal(x[f"{t}_geom_{key}"][g], x[f"{t}_geom_{key}"][g0], equal_nan=True)
    csr, csi = x[f"{t}_geom_solref"][g0], x[f"{t}_geom_solimp"][g0]
    lsr, lsi = x[f"{t}_joint_solreflimit"][aj[0]], x[f"{t}_joint_solimplimit"][aj[0]]
    assert _or(csr[1], 1.0) == 1.0 and _or(lsr[1], 1.0) == 1.0  # damping ratio 1
    out += [_or(x[f"{t}_geom_margin"][g0], 0.0), _or(csr[0], SOLREF_TC), *[_or(csi[i], SOLIMP[i]) for i in range(3)],
            _or(lsr[0], SOLREF_TC), *[_or(lsi[i], SOLIMP[i]) for i in range(3)]]
    cr = x[f"{t}_act_ctrlrange"]
    assert (cr == cr[0]).all()
    rootz = [j for j in range(len(jb)) if not hinge[j] and x[f"{t}_joint_axis"][j][2] == 1][0]
    out += [cr[0, 0], cr[0, 1], _or(x[f"{t}_joint_ref"][rootz], 0.0), x[f"{t}_joint_axis"][aj[0]][1]]
    return np.array(out, dtype=np.float64)


def sort_ends(v, nb, ng):
    """canonical order of each capsule's two end spheres inside an emei_model_constants vector"""
    v = v.copy()
    for k in range(ng):
        o = 1 + 6 * nb + 7 * k + 1
        e = v[o:o + 4].reshape(2, 2)
        v[o:o + 4] = e[np.lexsort((e[:, 1], e[:, 0]))].ravel()
    return v


def ip_expected(x):
    t = "ip"
    names = list(x["ip_geom_names"])
    cart, pole = names.index("cart"), names.index("cpole")
    ft = x["ip_geom_fromto"][pole]
    d = ft[3:] - ft[:3]
    length = np.linalg.norm(d)
    rp, half = x["ip_geom_size"][pole, 0], length / 2
    slider, = [j for j in range(2) if not x["ip_joint_is_hinge"][j]]
    assert x["ip_joint_limited"][slider] and np.isnan(x["ip_joint_solreflimit"]).all() and np.isnan(x["ip_geom_solref"]).all()
    assert all(_or(v, 0.0) == 0.0 for key in ("damping", "armature", "stiffness") for v in x[f"ip_joint_{key}"])
    return np.array([-x["ip_gravity"][2], capsule_mass(*x["ip_geom_size"][cart, :2]), capsule_mass(rp, half), capsule_inertia_perp(rp, half),
                     half, np.arctan2(d[0], d[2]), x["ip_act_gear"][0], *x["ip_act_ctrlrange"][0], *x["ip_joint_range"][slider],
                     SOLREF_TC, *SOLIMP, *(x["ip_joint_range"][1 - slider] * (np.pi / 180 if x["ip_angle_degree"] else 1.0))])


def dp_expected(x):
    names = list(x["dp_geom_names"])
    cart, p1, p2 = names.index("cart"), names.index("cpole"), names.index("cpole2")
    assert np.array_equal(x["dp_geom_fromto"][p1], x["dp_geom_fromto"][p2]) and x["dp_geom_size"][p1, 0] == x["dp_geom_size"][p2, 0]
    ft = x["dp_geom_fromto"][p1]
    assert ft[0] == ft[1] == ft[3] == ft[4] == 0 and ft[2] == 0  # along +z from the hinge
    half, rp = ft[5] / 2, x["dp_geom_size"][p1, 0]
    slider, = [j for j in range(3) if not x["dp_joint_is_hinge"][j]]
    L1 = x["dp_body_pos"][list(x["dp_body_names"]).index("pole2")][2]
    assert all(_or(v, 0.0) == 0.0 for key in ("damping", "armature", "stiffness") for v in x[f"dp_joint_{key}"])
    return np.array([x["dp_gravity"][0], -x["dp_gravity"][2], capsule_mass(*x["dp_geom_size"][cart, :2]), capsule_mass(rp, half),
                     capsule_inertia_perp(rp, half), half, L1, x["dp_act_gear"][0], *x["dp_act_ctrlrange"][0], *x["dp_joint_range"][slider],
                     x["dp_joint_margin"][slider], SOLREF_TC, *SOLIMP])


def _close(got, want, rtol=1e-12):
    assert got.shape == want.shape, (got.shape, want.shape)
    bad = np.nonzero(np.abs(got - want) > rtol * np.maximum(np.abs(want), 1.0))[0]
    assert bad.size == 0, [(int(i), float(got[i]), float(want[i])) for i in bad[:8]]


@pytest.mark.parametrize("model,tag,env_id,nb,ng", [("cheetah", "ch", 6, 7, 8), ("hopper", "hp", 11, 4, 4)])
def test_planar_bodies_match_the_xml(xml, model, tag, env_id, nb, ng):
    from oracle import oracle as O

    want = planar_expected(xml, tag)
    assert len(want) == 1 + 6 * nb + 7 * ng + 6 * len(xml[f"{tag}_act_joint"]) + 13
    _close(sort_ends(O.xml_constants(model), nb, ng), want)
    _close(sort_ends(kernel_constants(env_id), nb, ng), want)


def test_inverted_pendulum_matches_the_xml(xml):
    from oracle import oracle as O

    want = ip_expected(xml)
    _close(O.xml_constants("ip"), want)
    for env_id in (2, 3, 4, 5):
        _close(kernel_constants(env_id), want)
    # SwingUp's _update_model turns the pole body by pi and frees the hinge (inverted_pendulum.py:135-137); the Balancing
    # variants keep the file's hinge range of +-90 degrees: a second limit row since round 3 (oracle: ip_model.th_lo / th_hi)
    hinge, = [j for j in range(2) if xml["ip_joint_is_hinge"][j]]
    assert list(xml["ip_joint_range"][hinge]) == [-90.0, 90.0] and xml["ip_angle_degree"]
    m = O.ip_model()
    assert (m.th_lo, m.th_hi) == (-np.pi / 2, np.pi / 2)


def test_inverted_double_pendulum_matches_the_xml(xml):
    from oracle import oracle as O

    want = dp_expected(xml)
    _close(O.xml_constants("dp"), want)
    for env_id in (7, 8, 9, 10):
        _close(kernel_constants(env_id), want)


def test_getter_argument_checks():
    from emei_amd import _lib as L

    buf = (C.c_double * 4)()
    assert L.lib().emei_model_constants(6, C.cast(buf, C.c_void_p), 4) == L.ERR_INVALID  # capacity too small
    assert L.lib().emei_model_constants(0, C.cast(buf, C.c_void_p), 4) == L.ERR_UNSUPPORTED  # CartPole: constants are cartpole.py literals
    assert L.lib().emei_model_constants(99, C.cast(buf, C.c_void_p), 4) == L.ERR_INVALID
