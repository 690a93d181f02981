"""Pins against pieces of the reference that compile from their own sources with the image
libraries this image has (`make -C oracle ref` -> oracle/_ref/, git-ignored):

* `libiio_ref.so` = lib/iio/iio.c, the I/O library every reference tool reads and writes its
  frames, flows and masks with (src/main-flt.c:217-331, lib/tvl1flow/main.c) -> the file
  formats of host/imgio.c, in both directions;
* `awgn` = lib/imscript-lite/src/awgn.c -> the synthetic-noise generator of synth.py
  ("AWGN sigma with the reference generator").

Skipped when oracle/_ref is absent (no /root/reference to build it from)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
BIN = os.path.join(ROOT, "bwd-nlkalman_amd", "bin")


@pytest.fixture(scope="module")
def iio():
    so = os.path.join(REF, "libiio_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libiio_ref.so not built (needs /root/reference: make -C oracle ref)")
    L = ctypes.CDLL(so)
    fp, ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
    L.iio_read_image_float_vec.restype = fp
    L.iio_read_image_float_vec.argtypes = [ctypes.c_char_p, ip, ip, ip]
    L.iio_write_image_float_vec.argtypes = [ctypes.c_char_p, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.iio_write_image_uint8_vec.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint8),
                                            ctypes.c_int, ctypes.c_int, ctypes.c_int]

    class Iio:
        @staticmethod
        def read(path):
            w, h, pd = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            p = L.iio_read_image_float_vec(str(path).encode(), w, h, pd)
            assert p, path
            return np.ctypeslib.as_array(p, (h.value, w.value, pd.value)).copy()

        @staticmethod
        def write(path, a):
            a = np.ascontiguousarray(a, np.float32)
            a3 = a.reshape(a.shape[0], a.shape[1], -1)
            L.iio_write_image_float_vec(str(path).encode(), a3.ctypes.data_as(fp), a3.shape[1], a3.shape[0],
                                        a3.shape[2])

        @staticmethod
        def write_u8(path, a):
            a3 = np.ascontiguousarray(a, np.uint8).reshape(a.shape[0], a.shape[1], -1)
            L.iio_write_image_uint8_vec(str(path).encode(), a3.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                                        a3.shape[1], a3.shape[0], a3.shape[2])
    return Iio


@pytest.fixture(scope="module")
def conv(built):
    if not os.path.exists(os.path.join(BIN, "nlk-imgconv")):
        built.build()

    def run(src, dst):
        r = subprocess.run([os.path.join(BIN, "nlk-imgconv"), str(src), str(dst)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    return run


IMAGES = {
    "float_rgb": lambda r: r.uniform(-50, 300, (23, 31, 3)).astype(np.float32),
    "float_gray": lambda r: r.uniform(0, 255, (17, 29, 1)).astype(np.float32),
    "float_big_rgb": lambda r: r.normal(128, 60, (96, 131, 3)).astype(np.float32),   # several LZW strips
    "bytes_rgb": lambda r: r.integers(0, 256, (19, 27, 3)).astype(np.float32),       # stored as 8 bits
    "bytes_gray": lambda r: r.integers(0, 256, (21, 33, 1)).astype(np.float32),
    "nan_holes": lambda r: np.where(r.uniform(size=(16, 20, 3)) > 0.8, np.nan,
                                    r.uniform(0, 255, (16, 20, 3))).astype(np.float32),  # warped frames
}


@pytest.mark.parametrize("ext", ["tif", "tiff", "pfm"])
@pytest.mark.parametrize("name", sorted(IMAGES))
def test_frames_written_by_the_reference_library_read_identically(iio, conv, tmp_path, name, ext):
    a = IMAGES[name](np.random.default_rng(1))
    iio.write(tmp_path / f"ref.{ext}", a)
    back = iio.read(tmp_path / f"ref.{ext}")                 # what the reference's tools would see
    conv(tmp_path / f"ref.{ext}", tmp_path / "ours.pfm")
    ours = iio.read(tmp_path / "ours.pfm")
    assert ours.shape == back.shape
    assert np.array_equal(ours, back, equal_nan=True)
    assert np.array_equal(back, a, equal_nan=True)


@pytest.mark.parametrize("ext", ["tif", "pfm"])
@pytest.mark.parametrize("name", sorted(IMAGES))
def test_frames_we_write_are_read_identically_by_the_reference_library(iio, conv, tmp_path, name, ext):
    a = IMAGES[name](np.random.default_rng(2))
    iio.write(tmp_path / "in.pfm", a)
    conv(tmp_path / "in.pfm", tmp_path / f"ours.{ext}")
    back = iio.read(tmp_path / f"ours.{ext}")
    assert back.shape == a.shape and np.array_equal(back, a, equal_nan=True)


def test_pgm_and_ppm_files_read_like_the_reference_library_reads_them(iio, conv, tmp_path):
    """lib/iio/iio.c:1759-1805 (P2 / P3 ASCII, P5 / P6 binary with one or two bytes per sample, comments in the
    header, maxval not applied): host/imgio.c must return the same floats."""
    rng = np.random.default_rng(12)
    files = {}
    rgb8 = rng.integers(0, 256, (13, 17, 3), dtype=np.uint8)
    files["a.ppm"] = b"P6\n17 13\n255\n" + rgb8.tobytes()
    g8 = rng.integers(0, 200, (9, 21), dtype=np.uint8)
    files["b.pgm"] = b"P5 # a comment\n21\n# another\n9 199\n" + g8.tobytes()
    g16 = rng.integers(0, 65536, (7, 5)).astype(">u2")
    files["c.pgm"] = b"P5\n5 7\n65535\n" + g16.tobytes()
    c16 = rng.integers(0, 1024, (4, 6, 3)).astype(">u2")
    files["d.ppm"] = b"P6\n6 4\n1023\n" + c16.tobytes()
    files["e.pgm"] = ("P2\n# ascii\n4 3\n15\n" + " ".join(str(v) for v in rng.integers(0, 16, 12)) + "\n").encode()
    files["f.ppm"] = ("P3\n2 2\n255\n" + "\n".join("%g" % v for v in rng.uniform(-5, 300, 12)) + "\n").encode()
    for name, data in files.items():
        with open(tmp_path / name, "wb") as f:
            f.write(data)
        conv(tmp_path / name, tmp_path / (name + ".pfm"))
        ours, theirs = iio.read(tmp_path / (name + ".pfm")), iio.read(tmp_path / name)
        assert ours.shape == theirs.shape and np.array_equal(ours, theirs), name


def test_jpeg_files_read_like_the_reference_library_reads_them(iio, conv, tmp_path):
    """lib/iio/iio.c:1416-1460 reads JPEG through libjpeg (here IJG 9): baseline and progressive files must come back bit for bit -
    grey, full-resolution colour, and subsampled chroma (4:2:2, 4:2:0, and 4 x 1 / 1 x 4 made by patching a frame
    header), which libjpeg 9 brings to full resolution inside its fixed-point inverse transform (a 16-point
    transform of the 8 coefficients: host/imgio_jpeg.c) - with restart markers and optimised Huffman tables too."""
    from PIL import Image
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:67, 0:93]
    base = np.stack([128 + 100 * np.sin(xx / 7.) * np.cos(yy / 5.), 128 + 90 * np.cos(xx / 11. + yy / 3.), 60 + xx * 1.5], -1)
    img = np.clip(base + rng.normal(0, 12, base.shape), 0, 255).astype(np.uint8)
    noise = rng.integers(0, 256, (64, 80, 3), dtype=np.uint8)
    files = {"gray_q90": (img[..., 0], dict(quality=90)), "rgb444_q85": (img, dict(quality=85, subsampling=0)),
             "rgb444_q30": (img, dict(quality=30, subsampling=0)), "rgb444_q100": (img, dict(quality=100, subsampling=0)),
             "rgb444_opt": (img, dict(quality=75, subsampling=0, optimize=True)),
             "rgb444_rst": (img, dict(quality=80, subsampling=0, restart_marker_blocks=5)),
             "tiny": (img[:5, :3], dict(quality=90, subsampling=0)), "row": (img[:1, :, 0], dict(quality=95)),
             "rgb422_q85": (img, dict(quality=85, subsampling=1)), "rgb420_q85": (img, dict(quality=85, subsampling=2)),
             "rgb420_q50_rst": (img, dict(quality=50, subsampling=2, restart_marker_blocks=3)),
             "noise420_q95": (noise, dict(quality=95, subsampling=2)), "noise422_q30": (noise, dict(quality=30, subsampling=1)),
             # progressive: DC and AC bands in separate scans, successive approximation (refinement passes, end-of-band runs)
             "prog420": (img, dict(quality=85, subsampling=2, progressive=True)),
             "prog444": (img, dict(quality=90, subsampling=0, progressive=True)),
             "prog_gray": (img[..., 0], dict(quality=80, progressive=True)),
             "prog_noise": (noise, dict(quality=95, subsampling=2, progressive=True)),
             "prog422_q30": (noise, dict(quality=30, subsampling=1, progressive=True)),
             "prog_rst": (img, dict(quality=75, subsampling=2, progressive=True, restart_marker_blocks=2)),
             "prog_tiny": (img[:3, :5], dict(quality=90, progressive=True))}
    for name, (a, kw) in files.items():
        Image.fromarray(a).save(tmp_path / (name + ".jpg"), **kw)
    # luminance sampled 4 x 1 and 1 x 4: the 2 x 2 of a 4:2:0 file rewritten in its frame header (six blocks per MCU
    # either way: the picture is scrambled, the decoding is what is compared)
    Image.fromarray(img[:64, :].repeat(2, 1)[:, :96]).save(tmp_path / "mcu.jpg", quality=85, subsampling=2)   # 64 x 96: 24 MCUs
    files["mcu"] = None
    data = bytearray(open(tmp_path / "mcu.jpg", "rb").read())
    i = data.index(b"\xff\xc0")
    assert data[i + 11] == 0x22
    for hv in (0x41, 0x14):
        data[i + 11] = hv
        with open(tmp_path / ("hv%02x.jpg" % hv), "wb") as f:
            f.write(data)
        files["hv%02x" % hv] = None
    for name in files:
        conv(tmp_path / (name + ".jpg"), tmp_path / (name + ".pfm"))
        ours, theirs = iio.read(tmp_path / (name + ".pfm")), iio.read(tmp_path / (name + ".jpg"))
        assert ours.shape == theirs.shape and np.array_equal(ours, theirs), name


def test_flow_files_both_directions(iio, conv, tmp_path):
    fl = np.random.default_rng(3).normal(0, 3, (13, 22, 2)).astype(np.float32)
    iio.write(tmp_path / "ref.flo", fl)
    conv(tmp_path / "ref.flo", tmp_path / "ours.flo")
    assert open(tmp_path / "ref.flo", "rb").read() == open(tmp_path / "ours.flo", "rb").read()
    assert np.array_equal(iio.read(tmp_path / "ours.flo"), fl)


def test_occlusion_mask_png_both_directions(iio, conv, tmp_path):
    """plambda writes the mask as an 8-bit PNG 0 / 255 (scripts/nlkalman-seq.sh:70-72)."""
    m = (np.random.default_rng(4).uniform(size=(21, 33, 1)) > 0.7).astype(np.uint8) * 255
    iio.write_u8(tmp_path / "ref.png", m)
    conv(tmp_path / "ref.png", tmp_path / "ours.pfm")
    assert np.array_equal(iio.read(tmp_path / "ours.pfm"), m.astype(np.float32))
    conv(tmp_path / "ours.pfm", tmp_path / "ours.png")
    assert np.array_equal(iio.read(tmp_path / "ours.png"), m.astype(np.float32))
    rgb = np.random.default_rng(5).integers(0, 256, (14, 18, 3), dtype=np.uint8)
    iio.write_u8(tmp_path / "rgb.png", rgb)
    conv(tmp_path / "rgb.png", tmp_path / "rgb.pfm")
    assert np.array_equal(iio.read(tmp_path / "rgb.pfm"), rgb.astype(np.float32))


@pytest.mark.parametrize("shape,sigma,seed", [((37, 53, 3), 20.0, 1), ((64, 64, 1), 40.0, 0), ((20, 31, 3), 7.5, 12345)])
def test_synthetic_noise_equals_the_reference_awgn_tool(iio, tmp_path, shape, sigma, seed):
    import importlib
    synth = importlib.import_module("bwd-nlkalman_amd.synth")
    tool = os.path.join(REF, "awgn")
    if not os.path.exists(tool):
        pytest.skip("oracle/_ref/awgn not built")
    clean = synth.clean_frame(shape[1], shape[0], shape[2])
    iio.write(tmp_path / "clean.pfm", clean)
    env = dict(os.environ, SRAND=str(seed))
    r = subprocess.run([tool, repr(sigma), str(tmp_path / "clean.pfm"), str(tmp_path / "noisy.pfm")], env=env,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ref = iio.read(tmp_path / "noisy.pfm")
    ours = synth.awgn(clean, sigma, seed)
    assert np.array_equal(ours, ref)


# ---------------------------------------------------------------- the tools between the filter calls
def _tool(name):
    t = os.path.join(REF, name)
    if not os.path.exists(t):
        pytest.skip(f"oracle/_ref/{name} not built")
    return t


def _rflo(path):
    raw = open(path, "rb").read()
    assert raw[:4] == b"PIEH"
    w, h = np.frombuffer(raw[4:12], np.int32)
    return np.frombuffer(raw[12:], np.float32).reshape(h, w, 2)


MASK_EXPR = "x(0,0)[0] x(-1,0)[0] - x(0,0)[1] x(0,-1)[1] - + fabs {th} > 255 *"   # scripts/nlkalman-seq.sh:68-71


def colour_pair(w, h, seed):
    import importlib
    synth = importlib.import_module("bwd-nlkalman_amd.synth")
    n0, n1, _ = synth.noisy_pair(w, h, 3, 5.0, seed)
    return n0, n1


@pytest.mark.parametrize("w,h,args", [(72, 48, [8, 0, 0.40, 0, 0, 1]), (64, 80, []), (90, 60, [1, 0, 0.8, 0, 0, 0])])
def test_oracle_flow_equals_the_reference_tool_on_colour_files(iio, O, tmp_path, w, h, args):
    """`tvl1flow I0 I1 out nproc tau lambda theta nscales fscale` (lib/tvl1flow/main.c) on colour
    float TIFFs, as scripts/nlkalman-seq.sh:46-61 calls it: luminance conversion of the reader,
    parameter fall-backs, automatic number of scales and the flow itself."""
    tool = _tool("tvl1flow")
    c0, c1 = colour_pair(w, h, 7)
    iio.write(tmp_path / "a.tif", c0)
    iio.write(tmp_path / "b.tif", c1)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run([tool, str(tmp_path / "a.tif"), str(tmp_path / "b.tif"), str(tmp_path / "f.flo"),
                        *map(str, args)], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ref = _rflo(tmp_path / "f.flo")
    lam = args[2] if len(args) > 2 and args[2] > 0 else 0.15
    fscale = args[5] if len(args) > 5 else 0
    u, v = O.tvl1_flow(O.tvl1_gray(c0), O.tvl1_gray(c1), lam=lam, fscale=fscale)
    assert np.array_equal(ref[..., 0], u) and np.array_equal(ref[..., 1], v)


@pytest.mark.parametrize("th", [0.75, 0.25])
def test_oracle_occlusion_mask_equals_plambda(iio, O, tmp_path, th):
    tool = _tool("plambda")
    rng = np.random.default_rng(11)
    y, x = np.mgrid[0:40, 0:56].astype(np.float32)
    fl = np.stack([np.sin(x / 5) * 2 + rng.normal(0, .3, x.shape), np.cos(y / 4) * 2 + rng.normal(0, .3, x.shape)],
                  -1).astype(np.float32)
    iio.write(tmp_path / "f.flo", fl)
    r = subprocess.run([tool, str(tmp_path / "f.flo"), MASK_EXPR.format(th=th), "-o", str(tmp_path / "m.png")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ref = iio.read(tmp_path / "m.png")[..., 0]
    ours = O.tvl1_occlusion_mask(fl, th)
    assert 0.02 < (ref > 0).mean() < 0.9
    assert np.array_equal(ours, ref)


# ---------------------------------------------------------------- GPU: our tools beside the reference's
@pytest.mark.gpu
@pytest.mark.parametrize("w,h,args", [(72, 48, [8, 0, 0.40, 0, 0, 1]), (64, 80, []), (200, 120, [8, 0, 0.40, 0, 0, 1]),
                                      (90, 60, [1, 0, 0.8, 0, 0, 0])])
def test_gpu_tvl1flow_tool_writes_the_reference_tools_flow(iio, built, tmp_path, w, h, args):
    """bin/tvl1flow and the reference's tvl1flow on the same colour TIFFs with the same command
    line: the .flo files must be identical byte for byte."""
    ref_tool, ours = _tool("tvl1flow"), os.path.join(BIN, "tvl1flow")
    c0, c1 = colour_pair(w, h, 9)
    iio.write(tmp_path / "a.tif", c0)
    iio.write(tmp_path / "b.tif", c1)
    for tool, out in ((ref_tool, "ref.flo"), (ours, "ours.flo")):
        r = subprocess.run([tool, str(tmp_path / "a.tif"), str(tmp_path / "b.tif"), str(tmp_path / out),
                            *map(str, args)], env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    assert open(tmp_path / "ref.flo", "rb").read() == open(tmp_path / "ours.flo", "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("th", [0.75, 0.25])
def test_gpu_occlusion_mask_equals_plambda(iio, ctx, built, tmp_path, th):
    tool = _tool("plambda")
    rng = np.random.default_rng(12)
    h, w = 75, 131
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    fl = np.stack([np.sin(x / 5) * 2 + rng.normal(0, .3, x.shape), np.cos(y / 4) * 2 + rng.normal(0, .3, x.shape)],
                  -1).astype(np.float32)
    iio.write(tmp_path / "f.flo", fl)
    r = subprocess.run([tool, str(tmp_path / "f.flo"), MASK_EXPR.format(th=th), "-o", str(tmp_path / "m.png")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ref = iio.read(tmp_path / "m.png")[..., 0]
    d_fl, d_m = ctx.upload(fl), ctx.alloc(w * h * 4)
    ctx.occlusion_mask(d_m, d_fl, w, h, th)
    ours = ctx.download(d_m, (h, w))
    ctx.free(d_fl)
    ctx.free(d_m)
    assert np.array_equal(ours, ref)


@pytest.mark.gpu
def test_gpu_filter_tool_reads_and_writes_reference_files(iio, built, tmp_path):
    """nlkalman-flt fed with frames, a flow and a mask written by the reference's I/O library
    (float TIFF, .flo, 8-bit PNG) gives the same output as with PFM copies of the same data (up to the aggregation order), and
    the reference's reader gets back exactly what the tool computed."""
    rng = np.random.default_rng(13)
    import importlib
    synth = importlib.import_module("bwd-nlkalman_amd.synth")
    w, h = 96, 64
    n0, n1, _ = synth.noisy_pair(w, h, 3, 20.0, 5)
    fl = rng.normal(0, 0.4, (h, w, 2)).astype(np.float32)
    m = ((rng.uniform(size=(h, w, 1)) > 0.9) * 255).astype(np.uint8)
    for ext in ("tif", "pfm"):
        iio.write(tmp_path / f"n0.{ext}", n0)
        iio.write(tmp_path / f"n1.{ext}", n1)
    iio.write(tmp_path / "f.flo", fl)
    iio.write_u8(tmp_path / "m.png", m)
    iio.write(tmp_path / "m.pfm", m.astype(np.float32))
    flt = os.path.join(BIN, "nlkalman-flt")

    def run(*a):
        r = subprocess.run([flt, *map(str, a)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    outs = {}
    for ext, mask in (("tif", "m.png"), ("pfm", "m.pfm")):
        run("-i", tmp_path / f"n0.{ext}", "-s", 20, "--flt11", tmp_path / f"a0.{ext}", "--f2_p", 0)
        run("-i", tmp_path / f"n1.{ext}", "-s", 20, "-o", tmp_path / "f.flo", "-k", tmp_path / mask,
            "--flt10", tmp_path / f"a0.{ext}", "--flt11", tmp_path / f"a1.{ext}", "--f2_p", 0)
        outs[ext] = iio.read(tmp_path / f"a1.{ext}")
    assert outs["tif"].shape == (h, w, 3)
    # two runs of the filter agree to the float atomics' ordering noise only
    assert np.abs(outs["tif"] - outs["pfm"]).max() < 2e-3
    assert synth.psnr(outs["tif"], synth.clean_frame(w, h, 3, 1)) > synth.psnr(n1, synth.clean_frame(w, h, 3, 1)) + 4


# ------------------------------------------------------------ round 2: tiles, LZW writer, hostile files

def _fp_predict(t):
    """Adobe TIFF technical note 3 (predictor 3) on a block [rows, cols, comps] of float32: every row as byte planes,
    most significant bytes first, then differenced byte-wise with a stride of `comps`."""
    rows, cols, comps = t.shape
    be = t.astype(">f4").view(np.uint8).reshape(rows, cols * comps, 4)
    planes = np.ascontiguousarray(be.transpose(0, 2, 1)).reshape(rows, 4 * cols * comps)
    out = planes.copy()
    out[:, comps:] = planes[:, comps:] - planes[:, :-comps]
    return out.tobytes()


def _tiled_tiff(path, a, tw, tl, comp="raw", planar=False, big=False, fp_predictor=False):
    """A tiled TIFF assembled by hand from the TIFF 6.0 specification (section 15): tiles of
    tw x tl pixels, edge tiles padded, float32 samples; `comp` raw or deflate; optionally with the
    floating-point predictor (3)."""
    import struct
    import zlib
    h, w, ch = a.shape
    ax, ay = (w + tw - 1) // tw, (h + tl - 1) // tl
    tiles = []
    for pl in range(ch if planar else 1):
        for ty in range(ay):
            for tx in range(ax):
                t = np.zeros((tl, tw, 1 if planar else ch), np.float32)
                blk = a[ty * tl:(ty + 1) * tl, tx * tw:(tx + 1) * tw]
                blk = blk[..., pl:pl + 1] if planar else blk
                t[:blk.shape[0], :blk.shape[1]] = blk
                raw = _fp_predict(t) if fp_predictor else t.tobytes()
                tiles.append(zlib.compress(raw) if comp == "deflate" else raw)
    ents = [(256, 4, [w]), (257, 4, [h]), (258, 3, [32] * ch), (259, 3, [8 if comp == "deflate" else 1]),
            (262, 3, [2 if ch >= 3 else 1]), (277, 3, [ch]), (284, 3, [2 if planar else 1])] + \
           ([(317, 3, [3])] if fp_predictor else []) + \
           [(322, 4, [tw]), (323, 4, [tl]), (324, 16 if big else 4, None), (325, 16 if big else 4, [len(t) for t in tiles]),
            (339, 3, [3] * ch)]
    tsz = {3: 2, 4: 4, 16: 8}
    fsz, esz, hdr = (8, 20, 16) if big else (4, 12, 8)
    ifd_len = (8 if big else 2) + len(ents) * esz + fsz
    extra_off = hdr + ifd_len
    extra = b""
    # first pass: sizes of the out-of-line value arrays, then the tile offsets
    sizes = [tsz[ty] * (len(tiles) if v is None else len(v)) for _, ty, v in ents]
    data_off = extra_off + sum(s for s in sizes if s > fsz)
    offs, o = [], data_off
    for t in tiles:
        offs.append(o)
        o += len(t)
    body = b""
    for (tag, ty, v), s in zip(ents, sizes):
        v = offs if v is None else v
        fmt = {3: "H", 4: "I", 16: "Q"}[ty]
        packed = struct.pack("<" + fmt * len(v), *v)
        body += struct.pack("<HH", tag, ty) + struct.pack("<Q" if big else "<I", len(v))
        if s <= fsz:
            body += packed.ljust(fsz, b"\0")
        else:
            body += struct.pack("<Q" if big else "<I", extra_off + len(extra))
            extra += packed
    head = (struct.pack("<2sHHHQQ", b"II", 43, 8, 0, hdr, len(ents)) if big
            else struct.pack("<2sHIH", b"II", 42, hdr, len(ents)))
    with open(path, "wb") as f:
        f.write(head + body + b"\0" * fsz + extra + b"".join(tiles))


@pytest.mark.parametrize("comp,planar,big", [("raw", False, False), ("deflate", False, False), ("raw", True, False),
                                             ("deflate", False, True)])
def test_tiled_tiff_is_read_like_the_reference_library_reads_it(iio, conv, tmp_path, comp, planar, big):
    """lib/iio/iio.c:1463-1661 reads tiled files through libtiff; host/imgio.c decodes the tiles itself."""
    a = np.random.default_rng(8).normal(100, 50, (45, 70, 3)).astype(np.float32)
    _tiled_tiff(tmp_path / "t.tif", a, 32, 16, comp, planar, big)
    ref = iio.read(tmp_path / "t.tif")
    assert np.array_equal(ref, a)                      # the hand-made file is a valid tiled TIFF
    conv(tmp_path / "t.tif", tmp_path / "t.pfm")
    assert np.array_equal(iio.read(tmp_path / "t.pfm"), a)


@pytest.mark.parametrize("planar", [False, True])
def test_floating_point_predictor_tiff(iio, conv, tmp_path, planar):
    """TIFF predictor 3 (byte planes + byte-wise differencing; what e.g. GDAL and ImageMagick write for
    float images with Deflate): libtiff undoes it for the reference's reader (lib/iio/iio.c:1463-1661),
    host/imgio.c does it itself. The hand-made file must read back exactly through both."""
    a = np.random.default_rng(9).normal(100, 50, (37, 52, 3)).astype(np.float32)
    a[3, 5] = np.nan
    _tiled_tiff(tmp_path / "p3.tif", a, 32, 16, "deflate", planar, False, fp_predictor=True)
    assert np.array_equal(iio.read(tmp_path / "p3.tif"), a, equal_nan=True)   # libtiff agrees the file is valid
    conv(tmp_path / "p3.tif", tmp_path / "p3.pfm")
    assert np.array_equal(iio.read(tmp_path / "p3.pfm"), a, equal_nan=True)


@pytest.mark.parametrize("name", ["float_big_rgb", "bytes_rgb", "nan_holes", "float_gray"])
def test_lzw_tiff_we_write_is_read_by_the_reference_library(iio, tmp_path, name):
    """NLK_TIFF_LZW=1: the compression the reference's writer picks below 4 Mpixel (lib/iio/iio.c:3022-3026)."""
    a = IMAGES[name](np.random.default_rng(6))
    iio.write(tmp_path / "in.pfm", a)
    r = subprocess.run([os.path.join(BIN, "nlk-imgconv"), str(tmp_path / "in.pfm"), str(tmp_path / "o.tif")],
                       env=dict(os.environ, NLK_TIFF_LZW="1"), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    import struct
    raw = open(tmp_path / "o.tif", "rb").read()
    n = struct.unpack("<H", raw[8:10])[0]
    tags = {struct.unpack("<H", raw[10 + 12 * i:12 + 12 * i])[0]: struct.unpack("<H", raw[18 + 12 * i:20 + 12 * i])[0]
            for i in range(n)}
    assert tags[259] == 5                              # really LZW
    assert np.array_equal(iio.read(tmp_path / "o.tif"), a, equal_nan=True)
    # ... and by our own reader
    r = subprocess.run([os.path.join(BIN, "nlk-imgconv"), str(tmp_path / "o.tif"), str(tmp_path / "back.pfm")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(iio.read(tmp_path / "back.pfm"), a, equal_nan=True)


def test_lzw_long_runs_and_table_resets_round_trip(iio, tmp_path):
    """An image large and repetitive enough to fill the 12-bit table several times and to hit
    the KwKwK case, through our encoder -> the reference's decoder and our decoder."""
    rng = np.random.default_rng(7)
    a = np.repeat(rng.integers(0, 4, (64, 48, 3)), 8, axis=1).astype(np.float32)   # 64 x 384 x 3, long runs
    a[::7] = rng.integers(0, 256, a[::7].shape)
    iio.write(tmp_path / "in.pfm", a)
    env = dict(os.environ, NLK_TIFF_LZW="1")
    assert subprocess.run([os.path.join(BIN, "nlk-imgconv"), str(tmp_path / "in.pfm"), str(tmp_path / "o.tif")],
                          env=env).returncode == 0
    assert os.path.getsize(tmp_path / "o.tif") < a.size            # 8-bit samples, compressed
    assert np.array_equal(iio.read(tmp_path / "o.tif"), a)
    assert subprocess.run([os.path.join(BIN, "nlk-imgconv"), str(tmp_path / "o.tif"), str(tmp_path / "b.pfm")]).returncode == 0
    assert np.array_equal(iio.read(tmp_path / "b.pfm"), a)
