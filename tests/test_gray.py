"""The one-component (grayscale) extension, CPU side: the oracle's restatement against the committed fixtures and
against Pillow's decoder, and the product's parser with and without the opt-in flag.

PARITY UNPINNED: libKPEG cannot decode one-component files (its SOF0 parser reads three component triples whatever the
header says; the real reference answers TERMINATE on every fixture here, see test_reference_rejects).  The
extension applies the reference's own per-block arithmetic -- quirk Q1 included -- to the one component."""
import glob
import io
import json
import os

import numpy as np
import pytest

import kpeg_testlib as T

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MAN = json.load(open(os.path.join(GOLD, "manifest_gray.json")))


def _data(name):
    d = open(os.path.join(GOLD, name), "rb").read()
    assert T.sha256(d) == MAN[name]["jpg_sha256"]
    return d


@pytest.mark.parametrize("name", sorted(MAN))
def test_oracle_reproduces_the_committed_outputs(name):
    st, rgb = T.oracle_decode_gray(_data(name))
    assert st == T.DECODE_DONE
    assert rgb.shape == (MAN[name]["height"], MAN[name]["width"], 3)
    assert np.array_equal(rgb[..., 0], rgb[..., 1]) and np.array_equal(rgb[..., 0], rgb[..., 2])
    assert T.sha256(rgb.tobytes()) == MAN[name]["oracle_rgb_sha256"]


@pytest.mark.parametrize("name", sorted(MAN))
def test_oracle_agrees_with_an_independent_decoder(name):
    """Pillow (libjpeg) on the same file: within 2 levels (two different IDCTs) everywhere except the blocks the
    reference's quirk Q1 empties (coded DC difference of zero: the AC terms are dropped), whose number is pinned."""
    Image = pytest.importorskip("PIL.Image")
    d = _data(name)
    st, rgb = T.oracle_decode_gray(d)
    pil = np.asarray(Image.open(io.BytesIO(d)).convert("L")).astype(int)
    h, w = pil.shape
    blk = np.abs(rgb[..., 0].astype(int) - pil).reshape(h // 8, 8, w // 8, 8).max(axis=(1, 3))
    assert int((blk > 2).sum()) == MAN[name]["blocks_over_2_from_pillow"]
    assert (blk > 2).mean() < 0.06


def test_parser_needs_the_flag():
    import libkpeg_amd as K
    for name in sorted(MAN):
        d = _data(name)
        rst = "rst" in name
        rc, frame, scan = K.host_parse(d, allow_dri=rst)
        assert rc == K.TERMINATE and frame is None        # the reference's answer (test_reference_rejects)
        rc, frame, scan = K.host_parse(d, allow_dri=rst, allow_gray=True)
        assert rc == K.DECODE_DONE
        assert (frame.width, frame.height, frame.components) == (MAN[name]["width"], MAN[name]["height"], 1)
        # the one table pair stands in for both ids, as the device tables are indexed
        assert bytes(frame.dht[0][0].counts) == bytes(frame.dht[0][1].counts)
        assert bytes(frame.dht[1][0].symbols) == bytes(frame.dht[1][1].symbols)
        assert list(frame.qt[0]) == list(frame.qt[1])
        assert frame.restart_interval == (3 if rst else 0)


def test_three_component_files_are_untouched_by_the_flag():
    import libkpeg_amd as K
    d = open(os.path.join(GOLD, "pil_96x64_q85.jpg"), "rb").read()
    a = K.host_parse(d)
    b = K.host_parse(d, allow_gray=True)
    assert a[0] == b[0] == K.DECODE_DONE
    assert bytes(a[1])[:-4] == bytes(b[1])[:-4] and np.array_equal(a[2], b[2])
    assert b[1].components in (0, 3)


def test_gray_with_subsampling_factors_other_than_1x1_is_rejected():
    import libkpeg_amd as K
    d = bytearray(_data("gray_ramp_64x48_q75.jpg"))
    i = d.find(b"\xff\xc0")
    assert d[i + 11] == 0x11
    d[i + 11] = 0x22
    rc, _, _ = K.host_parse(bytes(d), allow_gray=True)
    assert rc != K.DECODE_DONE


@pytest.mark.skipif(not T.have_ref(), reason="real reference only exists in the build container")
def test_reference_rejects():
    for name in sorted(MAN):
        if "rst" in name:
            continue
        info, rgb = T.ref_decode(_data(name))
        assert info["status"] == "TERMINATE" and rgb is None
