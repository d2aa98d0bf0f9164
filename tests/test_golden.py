"""Committed golden fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py):
CPU: the oracle still reproduces them (regression pin of both flavours);
GPU (-m gpu): the HIP path reproduces the cv-flavour fixtures bit for bit and stays within the stated
tolerance (<= 1 LSB per channel, identical escape classification) of the glibc-flavour fixtures."""
import os
import sys

import numpy as np
import pytest

import common
import oracle_lib as O
import curvis_amd

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden as G  # noqa: E402

GOLD = os.path.dirname(os.path.abspath(G.__file__))


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


@pytest.mark.parametrize("name", sorted(G.BRUTE))
def test_oracle_reproduces_brute_fixtures(name):
    metric, res, pos, fwd, cap = G.BRUTE[name]
    g = load(name)
    sp, sn = common.make_skies(*G.SKY, "check")
    om, oc, _, _ = common.scene(metric, res=res, pos=pos, fwd=fwd)
    # the sincos-merged glibc flavours are held to the glibc fixture too: steps, codes, texels and pixels are the same
    for fl, tag in ((O.CV, "cv"), (O.LIBM, "libm"), (O.LIBM_SINCOS, "libm"), (O.LIBM_SINCOS_INL, "libm")):
        rgb, dbg, _ = O.render_image(fl, om, oc, O.sky(sp), O.sky(sn), cap, 100.0, 0.05, debug=True)
        assert np.array_equal(rgb, g["rgb_" + tag]) and np.array_equal(dbg["steps"], g["steps_" + tag])
        assert np.array_equal(dbg["code"], g["code_" + tag]) and np.array_equal(dbg["tx"], g["tx_" + tag])
        if tag == "cv":
            assert np.array_equal(common.bits(dbg["x"]), common.bits(g["x_cv"]))
            assert np.array_equal(common.bits(dbg["p"]), common.bits(g["p_cv"]))


@pytest.mark.parametrize("name", sorted(G.EFFICIENT))
def test_oracle_reproduces_efficient_fixtures(name):
    metric, res, pos, fwd, cap = G.EFFICIENT[name]
    g = load(name)
    sp, sn = common.make_skies(*G.SKY, "check")
    om, oc, _, _ = common.scene(metric, res=res, pos=pos, fwd=fwd)
    rgb, smp, _ = O.render_image_efficient(O.CV, om, oc, O.sky(sp), O.sky(sn), cap, 100.0, 0.05, 100, 100, 1e-5, 1e-5)
    assert np.array_equal(rgb, g["rgb_cv"])
    assert np.array_equal(common.bits(smp["a"]), common.bits(g["a_cv"]))
    assert np.array_equal(common.bits(smp["e"]), common.bits(g["e_cv"]))
    assert [smp["calls"], smp["steps"]] == list(g["calls_cv"])


def test_flavours_of_the_fixtures_agree():
    """what separates the two flavours is the last bit of six elementary functions; on every fixture that changes NO
    step count, escape code, texel index or pixel (the measured state of affairs, asserted as such; final states do
    differ in their last bits)"""
    for name in G.BRUTE:
        g = load(name)
        for f in ("steps", "code", "tx", "ty", "rgb"):
            assert np.array_equal(g[f + "_cv"], g[f + "_libm"]), (name, f)
    for name in G.EFFICIENT:
        g = load(name)
        assert len(g["a_cv"]) == len(g["a_libm"]) and list(g["calls_cv"]) == list(g["calls_libm"])
        assert np.abs(g["e_cv"] - g["e_libm"]).max() < 1e-9
        assert np.array_equal(g["rgb_cv"], g["rgb_libm"]), name


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(G.BRUTE))
def test_gpu_reproduces_brute_fixtures(gpu_ctx, name):
    metric, res, pos, fwd, cap = G.BRUTE[name]
    g = load(name)
    sp, sn = common.make_skies(*G.SKY, "check")
    _, _, pm, pc = common.scene(metric, res=res, pos=pos, fwd=fwd)
    sys_ = curvis_amd.RelativisticSystem(pm, curvis_amd.SphericalImage(sp), curvis_amd.SphericalImage(sn), pc,
                                         context=gpu_ctx)
    rgb, dbg = sys_.render_image_debug(cap, 100.0, 0.05)
    assert np.array_equal(rgb, g["rgb_cv"]) and np.array_equal(sys_.render_image(cap, 100.0, 0.05), g["rgb_cv"])
    assert np.array_equal(dbg["steps"], g["steps_cv"]) and np.array_equal(dbg["code"], g["code_cv"])
    assert np.array_equal(dbg["tx"], g["tx_cv"]) and np.array_equal(dbg["ty"], g["ty_cv"])
    assert np.array_equal(common.bits(dbg["x"])[..., 1:], common.bits(g["x_cv"])[..., 1:])
    assert np.array_equal(common.bits(dbg["p"]), common.bits(g["p_cv"]))
    # glibc flavour (the reference's arithmetic): every ray has the same step count, escape code and pixel
    # (measured on these fixtures and at full size: profiles/round2_libm_parity.txt)
    assert np.array_equal(dbg["steps"], g["steps_libm"]) and np.array_equal(dbg["code"], g["code_libm"])
    assert np.array_equal(rgb, g["rgb_libm"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(G.EFFICIENT))
def test_gpu_reproduces_efficient_fixtures(gpu_ctx, name):
    metric, res, pos, fwd, cap = G.EFFICIENT[name]
    g = load(name)
    sp, sn = common.make_skies(*G.SKY, "check")
    _, _, pm, pc = common.scene(metric, res=res, pos=pos, fwd=fwd)
    sys_ = curvis_amd.RelativisticSystem(pm, curvis_amd.SphericalImage(sp), curvis_amd.SphericalImage(sn), pc,
                                         context=gpu_ctx)
    rgb = sys_.render_image_efficient(cap, 100.0, 0.05, 100, 100, 1e-5, 1e-5)
    a, e, s = gpu_ctx.samples(0)
    assert np.array_equal(rgb, g["rgb_cv"])
    assert np.array_equal(common.bits(a), common.bits(g["a_cv"])) and np.array_equal(common.bits(e), common.bits(g["e_cv"]))
    assert np.array_equal(s, g["s_cv"])
    assert np.array_equal(rgb, g["rgb_libm"])   # pixel-identical to the glibc flavour (measured)
