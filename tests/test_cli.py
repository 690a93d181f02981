"""Command-line surface: flag grammar and messages of nlkalman-flt / nlkalman-smo
(reference: src/main-flt.c:71-149, src/main-smo.c:53-96, lib/argparse), the image
I/O they rely on, and (GPU) the whole file-in / file-out pipeline of
scripts/nlkalman-seq.sh on two frames against the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bwd-nlkalman_amd", "bin")
GOLD = os.path.join(ROOT, "tests", "golden")


def run(tool, *args, **kw):
    return subprocess.run([os.path.join(BIN, tool), *map(str, args)], capture_output=True, text=True, **kw)


@pytest.fixture(scope="module")
def tools(built):
    if not os.path.exists(os.path.join(BIN, "nlkalman-flt")):
        built.build()
    return BIN


def wpfm(path, a):
    a = np.ascontiguousarray(a, np.float32)
    h, w = a.shape[:2]
    ch = 1 if a.ndim == 2 else a.shape[2]
    with open(path, "wb") as f:
        f.write(b"%s\n%d %d\n-1.0\n" % (b"PF" if ch == 3 else b"Pf", w, h))
        f.write(a.tobytes())


def rpfm(path):
    with open(path, "rb") as f:
        t = f.readline().strip()
        w, h = map(int, f.readline().split())
        f.readline()
        return np.frombuffer(f.read(), np.float32).reshape(h, w, 3 if t == b"PF" else 1).copy()


def wflo(path, fl):
    h, w = fl.shape[:2]
    with open(path, "wb") as f:
        f.write(b"PIEH" + struct.pack("<ii", w, h) + np.ascontiguousarray(fl, np.float32).tobytes())


# ------------------------------------------------------------ CPU: grammar

def test_help_and_usage(tools):
    r = run("nlkalman-flt", "-h")
    assert r.returncode == 0 and r.stdout.startswith("Usage: nlkalman-flt [options] [[--] args]\n   or: nlkalman-flt [options]\n")
    for flag in ("-i, --nisy=<str>", "-o, --bflo=<str>", "-k, --bocc=<str>", "--flt10=<str>", "--flt20=<str>",
                 "--flt11=<str>", "--flt21=<str>", "-s, --sigma=<flt>", "--f1_p=<int>", "--f1_sx=<int>",
                 "--f1_st=<int>", "--f1_nx=<int>", "--f1_nt=<int>", "--f1_nt_agg=<int>", "--f1_bx=<flt>",
                 "--f1_bt=<flt>", "--f1_l=<flt>", "--f2_p=<int>", "--f2_nt_agg=<int>", "-v, --verbose=<int>"):
        assert flag in r.stdout, flag
    r = run("nlkalman-smo", "--help")
    assert r.returncode == 0
    for flag in ("--flt1=<str>", "--smo0=<str>", "-o, --fflo=<str>", "-k, --focc=<str>", "--smo1=<str>",
                 "--s1_p=<int>", "--s1_st=<int>", "--s1_nt=<int>", "--s1_nt_agg=<int>", "--s1_bt=<flt>", "--s1_l=<flt>"):
        assert flag in r.stdout, flag


def test_sequence_tool_usage_and_missing_frames(tools, tmp_path):
    """bin/nlkalman-seq (one-process scripts/nlkalman-seq.sh): argument errors are reported
    before any GPU work, with the script's own message for a missing frame (:19-28)."""
    r = run("nlkalman-seq")
    assert r.returncode == 1 and "SEQ FFR LFR SIG OUT [STP [FPM [SPM [OPM]]]]" in r.stderr
    r = run("nlkalman-seq", tmp_path / "%03d.tif", 1, 3, 20, tmp_path / "out")
    assert r.returncode == 1 and r.stdout.startswith("ERROR: ") and "001.tif not found" in r.stdout
    r = run("nlkalman-seq", tmp_path / "%03d.tif", 1, 3, 20, tmp_path / "out", 1, "", "", "1 2 3")
    assert r.returncode == 1 and "OPM must hold 6 numbers" in r.stderr
    r = run("nlkalman-seq", tmp_path / "%03d.tif", 1, 3, 20, tmp_path / "out", 1, "--f9_p 3")
    assert r.returncode == 1


def test_error_paths(tools):
    r = run("nlkalman-flt", "--bogus")
    assert r.returncode == 1 and "unknown option `--bogus`" in r.stderr and "Usage:" in r.stdout
    r = run("nlkalman-flt", "--f1_p", "0")
    assert r.returncode == 1 and "nothing to do" in r.stderr
    r = run("nlkalman-flt", "--f1_p=0", "--flt21", "x.tif")
    assert r.returncode == 1 and "f1_p == 0 and no input path given" in r.stderr
    r = run("nlkalman-flt", "-s", "20")
    assert r.returncode == 1 and "no output path given" in r.stderr
    r = run("nlkalman-flt", "-s", "abc")
    assert r.returncode == 1 and "expects a numerical value" in r.stderr
    r = run("nlkalman-flt", "--f1_p", "1.5")
    assert r.returncode == 1 and "expects an integer value" in r.stderr
    r = run("nlkalman-flt", "--sigma")
    assert r.returncode == 1 and "requires a value" in r.stderr
    r = run("nlkalman-flt", "-i", "/nonexistent.tif", "--flt11", "o.tif", "-s20")
    assert r.returncode == 1 and "Error while openning" in r.stderr
    r = run("nlkalman-smo", "-s", "20")
    assert r.returncode == 1 and "no output path given" in r.stderr
    r = run("nlkalman-smo", "--smo1", "o.tif", "--s1_p", "0")
    assert r.returncode == 1 and "s1_p == 0" in r.stderr


def test_verbose_prints_resolved_defaults(tools, tmp_path):
    """-v 1 dumps the parameters after nlkalman_default_params (reference: src/main-flt.c:156-212);
    integers accept strtol base prefixes, the last occurrence of a flag wins."""
    r = run("nlkalman-flt", "-i", "/nonexistent.pfm", "--flt11", "a.tif", "--flt21", "b.tif", "-s", "40",
            "--f1_p", "12", "--f1_p", "0x8", "--f2_nt=7", "-v1")
    out = r.stdout
    assert "noise         40.00" in out
    assert "first filtering parameters:\n\tpatch      8\n\tsearch_x   10\n\tsearch_t   5\n\tnp_x       60\n\tnp_t       30\n\tnp_tagg    20\n" in out
    assert "second filtering parameters:" in out and "\tnp_t       7\n\tnp_tagg    1\n" in out
    assert "beta_t     1.85" in out and "beta_x     0.37" in out


# ------------------------------------------------------------ CPU: image I/O

def test_image_io_against_pil(tools, tmp_path):
    from PIL import Image
    rng = np.random.default_rng(0)
    a = rng.uniform(-50, 300, (23, 31, 3)).astype(np.float32)
    wpfm(tmp_path / "a.pfm", a)
    for ext in ("tif", "pfm"):
        assert run("nlk-imgconv", tmp_path / "a.pfm", tmp_path / f"b.{ext}").returncode == 0
    assert np.array_equal(rpfm(tmp_path / "b.pfm"), a)
    # our float TIFF is readable by an independent decoder, channel by channel
    run("nlk-imgconv", tmp_path / "b.tif", tmp_path / "c.pfm")
    assert np.array_equal(rpfm(tmp_path / "c.pfm"), a)
    g = rng.uniform(0, 255, (17, 29)).astype(np.float32)
    wpfm(tmp_path / "g.pfm", g)
    run("nlk-imgconv", tmp_path / "g.pfm", tmp_path / "g.tif")
    assert np.array_equal(np.asarray(Image.open(tmp_path / "g.tif"), np.float32), g)
    # TIFFs written by another encoder: float LZW / deflate / uncompressed, 8- and 16-bit
    for comp in ("raw", "tiff_lzw", "tiff_adobe_deflate"):
        Image.fromarray(g).save(tmp_path / "p.tif", compression=comp)
        assert run("nlk-imgconv", tmp_path / "p.tif", tmp_path / "p.pfm").returncode == 0
        assert np.array_equal(rpfm(tmp_path / "p.pfm")[..., 0], g), comp
    rgb8 = rng.integers(0, 256, (19, 27, 3), dtype=np.uint8)
    Image.fromarray(rgb8).save(tmp_path / "q.tif", compression="tiff_lzw")
    run("nlk-imgconv", tmp_path / "q.tif", tmp_path / "q.pfm")
    assert np.array_equal(rpfm(tmp_path / "q.pfm"), rgb8.astype(np.float32))
    # integer-valued [0,255] data is stored as 8 bits, like the reference's writer
    wpfm(tmp_path / "i.pfm", rgb8.astype(np.float32))
    run("nlk-imgconv", tmp_path / "i.pfm", tmp_path / "i.tif")
    im = Image.open(tmp_path / "i.tif")
    assert im.mode == "RGB" and np.array_equal(np.asarray(im), rgb8)
    # PNG: occlusion masks are 8-bit gray 0/255 (scripts/nlkalman-seq.sh:70-72)
    m = (rng.uniform(size=(21, 33)) > 0.7).astype(np.uint8) * 255
    Image.fromarray(m).save(tmp_path / "m.png")
    run("nlk-imgconv", tmp_path / "m.png", tmp_path / "m.pfm")
    assert np.array_equal(rpfm(tmp_path / "m.pfm")[..., 0], m.astype(np.float32))
    Image.fromarray(rgb8).save(tmp_path / "r.png")
    run("nlk-imgconv", tmp_path / "r.png", tmp_path / "r.pfm")
    assert np.array_equal(rpfm(tmp_path / "r.pfm"), rgb8.astype(np.float32))
    run("nlk-imgconv", tmp_path / "r.pfm", tmp_path / "w.png")
    assert np.array_equal(np.asarray(Image.open(tmp_path / "w.png")), rgb8)
    # PGM / PPM: binary with 8 and 16 bits per sample (another encoder's files), ASCII with comments in the header
    Image.fromarray(rgb8).save(tmp_path / "r.ppm")
    run("nlk-imgconv", tmp_path / "r.ppm", tmp_path / "r2.pfm")
    assert np.array_equal(rpfm(tmp_path / "r2.pfm"), rgb8.astype(np.float32))
    Image.fromarray(m).save(tmp_path / "m.pgm")
    run("nlk-imgconv", tmp_path / "m.pgm", tmp_path / "m2.pfm")
    assert np.array_equal(rpfm(tmp_path / "m2.pfm")[..., 0], m.astype(np.float32))
    g16 = rng.integers(0, 65536, (11, 14)).astype(np.uint16)
    with open(tmp_path / "g16.pgm", "wb") as f:
        f.write(b"P5\n# sixteen bits, most significant byte first\n14 11\n65535\n" + g16.astype(">u2").tobytes())
    run("nlk-imgconv", tmp_path / "g16.pgm", tmp_path / "g16.pfm")
    assert np.array_equal(rpfm(tmp_path / "g16.pfm")[..., 0], g16.astype(np.float32))
    with open(tmp_path / "t.ppm", "w") as f:
        f.write("P3 # ascii\n# width height\n2 2\n255\n1 2 3  4 5 6\n7 8 9\n10.5 -1 300\n")
    run("nlk-imgconv", tmp_path / "t.ppm", tmp_path / "t.pfm")
    assert np.array_equal(rpfm(tmp_path / "t.pfm").ravel(), np.float32([1, 2, 3, 4, 5, 6, 7, 8, 9, 10.5, -1, 300]))
    for bad in (b"P5\n4 4\n255\nxx", b"P6\n2 2\n70000\n" + bytes(12), b"P2\n2 2\n255\n1 2 3\n", b"P5\n-3 2\n255\n"):
        with open(tmp_path / "bad.pgm", "wb") as f:
            f.write(bad)
        assert run("nlk-imgconv", tmp_path / "bad.pgm", tmp_path / "x.pfm").returncode == 1
    # JPEG (baseline): another decoder's numbers for grey and full-resolution colour; what is not read fails loudly
    yy, xx = np.mgrid[0:40, 0:52]
    pic = np.clip(np.stack([128 + 90 * np.sin(xx / 6.), 100 + yy * 3., 128 + 80 * np.cos(yy / 4. + xx / 9.)], -1)
                  + rng.normal(0, 8, (40, 52, 3)), 0, 255).astype(np.uint8)
    for name, a, kw in (("j1", pic[..., 1], dict(quality=90)), ("j3", pic, dict(quality=85, subsampling=0))):
        Image.fromarray(a).save(tmp_path / (name + ".jpg"), **kw)
        assert run("nlk-imgconv", tmp_path / (name + ".jpg"), tmp_path / (name + ".pfm")).returncode == 0
        got = rpfm(tmp_path / (name + ".pfm"))
        want = np.asarray(Image.open(tmp_path / (name + ".jpg")), np.float32).reshape(got.shape)
        assert np.array_equal(got, want), name
    Image.fromarray(pic[..., 0]).save(tmp_path / "prog.jpg", progressive=True, quality=80)   # several scans, refinements
    assert run("nlk-imgconv", tmp_path / "prog.jpg", tmp_path / "prog.pfm").returncode == 0
    assert np.array_equal(rpfm(tmp_path / "prog.pfm")[..., 0], np.asarray(Image.open(tmp_path / "prog.jpg"), np.float32))
    data = open(tmp_path / "j3.jpg", "rb").read()
    for cut in (20, 200, len(data) // 2):
        with open(tmp_path / "cut.jpg", "wb") as f:
            f.write(data[:cut])
        assert run("nlk-imgconv", tmp_path / "cut.jpg", tmp_path / "x.pfm").returncode in (0, 1)   # (no crash)
    # flow files
    fl = rng.normal(0, 2, (9, 13, 2)).astype(np.float32)
    wflo(tmp_path / "f.flo", fl)
    run("nlk-imgconv", tmp_path / "f.flo", tmp_path / "f2.flo")
    assert open(tmp_path / "f.flo", "rb").read() == open(tmp_path / "f2.flo", "rb").read()
    assert run("nlk-imgconv", tmp_path / "nope.tif", tmp_path / "x.pfm").returncode == 1


# ------------------------------------------------------------ the tools behind a resident server

import contextlib
import time


@contextlib.contextmanager
def server(tmp_path, *flags, **more_env):
    """bin/nlk-server on a socket of its own; the tools find it through NLK_SERVER"""
    sock = str(tmp_path / "nlk.sock")
    proc = subprocess.Popen([os.path.join(BIN, "nlk-server"), *flags, sock], stdout=subprocess.PIPE,
                            stderr=subprocess.PIPE, env=dict(os.environ, **more_env))
    try:
        for _ in range(600):       # (with a device context: the ~0.3 s of a HIP start, once)
            if os.path.exists(sock) or proc.poll() is not None:
                break
            time.sleep(0.05)
        assert proc.poll() is None and os.path.exists(sock), proc.stderr.read()
        yield dict(os.environ, NLK_SERVER=sock)
    finally:
        subprocess.run([os.path.join(BIN, "nlk-server"), "--stop", sock], capture_output=True)
        try:
            proc.wait(timeout=20)
        except subprocess.TimeoutExpired:
            proc.kill()


def test_tools_behind_the_resident_server_speak_the_same_grammar(tools, tmp_path):
    """NLK_SERVER=<socket>: the tool hands its arguments, directory, stdout and stderr to bin/nlk-server and
    exits with the status it returns (host/cli_server.h). Usage text, messages, exit codes and where they are
    printed must be what the tool prints by itself (reference: src/main-flt.c:71-149, lib/argparse); without a
    listener the tool works by itself; the server survives tools that leave through exit()."""
    with server(tmp_path, "--lazy") as env:
        for tool, args in (("nlkalman-flt", ["-h"]), ("nlkalman-smo", ["--help"]), ("nlkalman-flt", ["--nope"]),
                           ("nlkalman-smo", ["-s"]), ("nlkalman-flt", ["--f1_p", "x"]), ("nlkalman-flt", ["--f1_p", "0"]),
                           ("nlkalman-smo", ["--s1_p", "0", "--smo1", "x.tif"]), ("tvl1flow", []), ("decompose", []),
                           ("recompose", ["-h"]), ("merge_coarse", ["a"]),
                           ("nlkalman-flt", ["-i", "missing.tif", "--flt11", "o.tif", "-s", "10"])):
            local = run(tool, *args, cwd=tmp_path)
            remote = run(tool, *args, cwd=tmp_path, env=env)
            assert (remote.returncode, remote.stdout, remote.stderr) == (local.returncode, local.stdout, local.stderr), \
                (tool, args)
        # clients that do not speak the protocol (wrong magic, absurd length, a connection that says nothing, no
        # descriptors attached) are dropped; the server goes on serving
        import socket
        import struct as st
        for junk in (b"", b"\x10\x00\x00\x00" + b"XXXX\0tool\0.\0" + b"0\0", st.pack("<I", 0x7fffffff) + b"abc",
                     st.pack("<I", 12) + b"NLK1\0nlkalman-flt\0"):
            c = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            c.connect(env["NLK_SERVER"])
            if junk:
                c.sendall(junk)
            c.close()
        r = run("nlkalman-flt", "-h", env=env)
        assert r.returncode == 0 and r.stdout.startswith("Usage: nlkalman-flt")
        #  the socket is this user's alone; a second server does not take a live one's socket over; a client
        # that connects and says nothing is dropped after the server's I/O timeout instead of holding every step up;
        # descriptors beyond the two a request carries are closed, not leaked
        import stat
        import time
        assert stat.S_IMODE(os.stat(env["NLK_SERVER"]).st_mode) == 0o600
        r2 = subprocess.run([os.path.join(BIN, "nlk-server"), "--lazy", env["NLK_SERVER"]], capture_output=True, text=True, timeout=60)
        assert r2.returncode != 0 and "another server is listening" in r2.stderr
        mute = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        mute.connect(env["NLK_SERVER"])               # ... and nothing more
        t0 = time.time()
        r = run("nlkalman-flt", "-h", env=env)         # served once the mute client has timed out (10 s)
        assert r.returncode == 0 and r.stdout.startswith("Usage: nlkalman-flt") and time.time() - t0 < 30
        mute.close()
        import array
        fat = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        fat.connect(env["NLK_SERVER"])
        fat.sendmsg([st.pack("<I", 12)], [(socket.SOL_SOCKET, socket.SCM_RIGHTS, array.array("i", [1, 2, 1, 2, 1]).tobytes())])
        fat.close()
        r = run("nlkalman-flt", "-h", env=env)
        assert r.returncode == 0 and r.stdout.startswith("Usage: nlkalman-flt")
        r = run("nlkalman-flt", "-h", env=dict(env, NLK_SERVER=str(tmp_path / "nobody.sock")))
        assert r.returncode == 0 and r.stdout.startswith("Usage: nlkalman-flt")     # no listener: by itself
    r = subprocess.run([os.path.join(BIN, "nlk-server"), "--stop", str(tmp_path / "nlk.sock")], capture_output=True,
                       text=True)
    assert r.returncode == 1 and "nobody listens" in r.stderr


@pytest.mark.gpu
def test_pipeline_through_the_resident_server(tools, tmp_path):
    """The steps scripts/nlkalman-seq.sh runs per frame (flow, first and second iteration, smoother) as separate
    processes - once by themselves, once with bin/nlk-server doing the work: the files written must be identical
    byte for byte (bit-reproducible aggregation on both sides: NLK_DETERMINISTIC), device memory must not pile up
    over requests, and a step must no longer cost a HIP start."""
    I = cases.inputs("rgb72x48_s40")
    S = "%g" % I["sigma"]
    wpfm(str(tmp_path / "n0.pfm"), I["n0"])
    wpfm(str(tmp_path / "n1.pfm"), I["n1"])
    wflo(str(tmp_path / "b.flo"), I["flow"])

    def steps(tag, env):
        q = lambda f: str(tmp_path / (tag + f))  # noqa: E731
        seq = [("tvl1flow", str(tmp_path / "n1.pfm"), str(tmp_path / "n0.pfm"), q("tv.flo")),
               ("nlkalman-flt", "-i", str(tmp_path / "n0.pfm"), "-s", S, "--flt11", q("f1_0.pfm"), "--flt21", q("f2_0.pfm")),
               ("nlkalman-flt", "-i", str(tmp_path / "n1.pfm"), "-s", S, "--f2_p", "0", "-o", str(tmp_path / "b.flo"),
                "--flt10", q("f1_0.pfm"), "--flt11", q("f1_1.pfm")),
               ("nlkalman-flt", "-i", str(tmp_path / "n1.pfm"), "-s", S, "--f1_p", "0", "-o", str(tmp_path / "b.flo"),
                "--flt11", q("f1_1.pfm"), "--flt20", q("f2_0.pfm"), "--flt21", q("f2_1.pfm")),
               ("nlkalman-smo", "--flt1", q("f2_0.pfm"), "--smo0", q("f2_1.pfm"), "--smo1", q("s1_0.pfm"), "-s", S),
               # (the multiscale wrapper's tools, scripts/msnlkalman-seq.sh:56-60, 106-110)
               ("decompose", str(tmp_path / "n0.pfm"), q("ms"), "2", ".pfm"),
               ("recompose", q("ms"), "2", ".pfm", q("rec.pfm"), "-c", "0.7")]
        t0 = time.perf_counter()
        for a in seq:
            r = run(*a, env=env)
            assert r.returncode == 0, (a, r.stderr)
        return (time.perf_counter() - t0) / len(seq)

    alone = steps("a_", dict(os.environ, NLK_DETERMINISTIC="1"))
    with server(tmp_path, NLK_DETERMINISTIC="1") as env:
        steps("w_", env)                      # (code objects load on first use)
        served = min(steps("s_", env) for _ in range(3))
        for _ in range(40):                   # 200 more requests: what a request allocates is released
            steps("s_", env)
    for f in ("tv.flo", "f1_0.pfm", "f2_0.pfm", "f1_1.pfm", "f2_1.pfm", "s1_0.pfm", "ms0.pfm", "ms1.pfm", "rec.pfm"):
        a, b = (open(str(tmp_path / (t + f)), "rb").read() for t in ("a_", "s_"))
        assert a == b, f
    assert served < 0.1 and served < alone / 2, (served, alone)


# ------------------------------------------------------------ GPU: pipeline

@pytest.mark.gpu
@pytest.mark.parametrize("name", ["rgb72x48_s40", "gray64_s20"])
def test_two_frame_pipeline_files(tools, O, tmp_path, name):
    """The call sequence of scripts/nlkalman-seq.sh (frame 0: flt1+flt2 spatial;
    frame 1: flt1 then flt2 as separate processes with flow + occlusion mask;
    then nlkalman-smo), with TIFF / PNG / FLO files as the pipelines use them."""
    from PIL import Image
    I = cases.inputs(name)
    S = "%g" % I["sigma"]
    p = lambda f: str(tmp_path / f)  # noqa: E731
    wpfm(p("n0.pfm"), I["n0"])
    run("nlk-imgconv", p("n0.pfm"), p("n0.tif"))
    wpfm(p("n1.pfm"), I["n1"])
    run("nlk-imgconv", p("n1.pfm"), p("n1.tif"))
    wflo(p("b.flo"), I["flow"])
    wflo(p("f.flo"), -I["flow"])
    Image.fromarray(I["occ"].astype(np.uint8)).save(p("occ.png"))
    r = run("nlkalman-flt", "-i", p("n0.tif"), "-s", S, "--flt11", p("f1_0.tif"), "--flt21", p("f2_0.tif"))
    assert r.returncode == 0, r.stderr
    r = run("nlkalman-flt", "-i", p("n1.tif"), "-s", S, "--f2_p", "0", "-o", p("b.flo"), "-k", p("occ.png"),
            "--flt10", p("f1_0.tif"), "--flt11", p("f1_1.tif"))
    assert r.returncode == 0, r.stderr
    r = run("nlkalman-flt", "-i", p("n1.tif"), "-s", S, "--f1_p", "0", "-o", p("b.flo"), "-k", p("occ.png"),
            "--flt11", p("f1_1.tif"), "--flt20", p("f2_0.tif"), "--flt21", p("f2_1.tif"))
    assert r.returncode == 0, r.stderr
    r = run("nlkalman-smo", "--flt1", p("f2_0.tif"), "--smo0", p("f2_1.tif"), "-o", p("f.flo"), "-k", p("occ.png"),
            "--smo1", p("s1_0.tif"), "-s", S)
    assert r.returncode == 0, r.stderr
    r = run("nlkalman-smo", "--flt1", p("f2_0.tif"), "--smo0", p("f2_1.tif"), "--smo1", p("s1_x.tif"), "-s", S,
            env=dict(os.environ, NLK_SMO_REFERENCE_EXIT="1"))
    assert r.returncode == 1  # the reference's (odd) success status, on request

    def load(f):
        run("nlk-imgconv", p(f), p(f + ".pfm"))
        return rpfm(p(f + ".pfm"))
    got = {k: load(k + ".tif") for k in ("f1_0", "f2_0", "f1_1", "f2_1", "s1_0")}
    # oracle, stage by stage on the files the tools actually exchanged
    s = I["sigma"]
    p1, p2, ps = (O.default_params(s, m) for m in (O.FLT1, O.FLT2, O.SMO1))
    o0, o1 = O.rgb2opp(I["n0"]), O.rgb2opp(I["n1"])
    f1_0 = O.filter_frame(o0, None, None, s, p1)
    cases.assert_close(got["f1_0"], O.opp2rgb(f1_0), "f1_0")
    cases.assert_close(got["f2_0"], O.opp2rgb(O.filter_frame(o0, None, f1_0, s, p2)), "f2_0")
    w1 = O.warp_bicubic(O.rgb2opp(got["f1_0"]), I["flow"], I["occ"])
    f1_1 = O.filter_frame(o1, w1, None, s, p1)
    cases.assert_close(got["f1_1"], O.opp2rgb(f1_1), "f1_1")
    w2 = O.warp_bicubic(O.rgb2opp(got["f2_0"]), I["flow"], I["occ"])
    f2_1 = O.filter_frame(o1, w2, O.rgb2opp(got["f1_1"]), s, p2)
    cases.assert_close(got["f2_1"], O.opp2rgb(f2_1), "f2_1")
    ws = O.warp_bicubic(O.rgb2opp(got["f2_1"]), -I["flow"], I["occ"])
    s1 = O.smooth_frame(O.rgb2opp(got["f2_0"]), ws, None, s, ps)
    cases.assert_close(got["s1_0"], O.opp2rgb(s1), "s1_0")
    # and against the survey build of the reference's own sources (supplementary evidence)
    with np.load(os.path.join(GOLD, "survey_shim_" + name + ".npz")) as g:
        for k in ("f1_0", "f2_0"):
            cases.assert_close(got[k], g[k], "survey " + k, maxabs=2e-3)
        clean = I["clean1"]
        assert abs(cases.synth.psnr(got["f2_1"], clean) - cases.synth.psnr(g["f2_1"], clean)) <= 0.02


@pytest.mark.gpu
def test_reference_mains_on_top_of_the_library(tools, tmp_path):
    """The reference's own main-flt.c / main-smo.c, compiled unmodified against include/nlkalman.h and linked against
    libnlkalman.so in the build container (oracle/Makefile: _ref/refmain-*; tests/test_host.py links them), run the
    script's call sequence on two frames beside this repo's own tools: the same files out up to the order of the frame
    path's float atomics (both front ends hand the same arrays to the same library; PFM files - those binaries carry
    iio without image libraries)."""
    ref = {t: os.path.join(ROOT, "oracle", "_ref", "refmain-nlkalman-" + t) for t in ("flt", "smo")}
    if not all(os.path.exists(x) for x in ref.values()):
        pytest.skip("oracle/_ref/refmain-* not built (make -C oracle ref, in the build container)")
    I = cases.inputs("rgb72x48_s40")
    S = "%g" % I["sigma"]
    wpfm(tmp_path / "n0.pfm", I["n0"])
    wpfm(tmp_path / "n1.pfm", I["n1"])
    wflo(tmp_path / "b.flo", I["flow"])
    wflo(tmp_path / "f.flo", -I["flow"])
    wpfm(tmp_path / "occ.pfm", I["occ"])

    def seq(tag, flt, smo):
        p = lambda f: str(tmp_path / f)          # noqa: E731
        o = lambda f: str(tmp_path / (tag + f))  # noqa: E731
        calls = [
            (flt, ["-i", p("n0.pfm"), "-s", S, "--flt11", o("f1_0.pfm"), "--flt21", o("f2_0.pfm")], 0),
            (flt, ["-i", p("n1.pfm"), "-s", S, "--f2_p", "0", "-o", p("b.flo"), "-k", p("occ.pfm"),
                   "--flt10", o("f1_0.pfm"), "--flt11", o("f1_1.pfm")], 0),
            (flt, ["-i", p("n1.pfm"), "-s", S, "--f1_p", "0", "-o", p("b.flo"), "-k", p("occ.pfm"),
                   "--flt11", o("f1_1.pfm"), "--flt20", o("f2_0.pfm"), "--flt21", o("f2_1.pfm")], 0),
            (smo, ["--flt1", o("f2_0.pfm"), "--smo0", o("f2_1.pfm"), "-o", p("f.flo"), "-k", p("occ.pfm"),
                   "--smo1", o("s1_0.pfm"), "-s", S], None),   # (the reference's smoother returns 1 on success)
        ]
        for exe, args, rc in calls:
            r = subprocess.run([exe, *args], capture_output=True, text=True)
            assert (r.returncode == rc) if rc is not None else (r.returncode in (0, 1)), (exe, r.stderr)
    seq("ours_", os.path.join(BIN, "nlkalman-flt"), os.path.join(BIN, "nlkalman-smo"))
    seq("ref_", ref["flt"], ref["smo"])
    for f in ("f1_0", "f2_0", "f1_1", "f2_1", "s1_0"):
        a, b = rpfm(tmp_path / ("ours_" + f + ".pfm")), rpfm(tmp_path / ("ref_" + f + ".pfm"))
        assert a.shape == b.shape and np.isfinite(a).all()
        # (two runs of the frame path differ by the order of its float atomics: FP noise, not bytes)
        cases.assert_close(a, b, "reference main vs this repo's tool, " + f)


def test_hostile_image_files_are_rejected(tools, tmp_path):
    """LZW codes beyond the next free entry, stale-table cycles, short
    uncompressed strips, the floating-point predictor on integer samples and absurd header sizes must end in an
    error message, never in a crash or a hang."""
    import struct
    from PIL import Image
    g = np.random.default_rng(0).uniform(0, 255, (17, 29)).astype(np.float32)
    Image.fromarray(g).save(tmp_path / "p.tif", compression="tiff_lzw")
    raw = bytearray(open(tmp_path / "p.tif", "rb").read())

    def entries(buf):
        off = struct.unpack("<I", buf[4:8])[0]
        n = struct.unpack("<H", buf[off:off + 2])[0]
        return {struct.unpack("<H", buf[off + 2 + 12 * i:off + 4 + 12 * i])[0]: off + 2 + 12 * i for i in range(n)}
    ent = entries(raw)
    strip_off = struct.unpack("<I", raw[ent[273] + 8:ent[273] + 12])[0]
    # (1) a stream whose second code is far beyond the table: clear, literal 65, code 400
    bad = bytearray(raw)
    bits = f"{256:09b}{65:09b}{400:09b}{257:09b}"
    bits += "0" * (-len(bits) % 8)
    evil = bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))
    bad[strip_off:strip_off + len(evil)] = evil
    open(tmp_path / "bad1.tif", "wb").write(bad)
    r = run("nlk-imgconv", tmp_path / "bad1.tif", tmp_path / "o.pfm")
    assert r.returncode != 0 and "LZW" in r.stderr
    # (2) random bytes as an LZW strip: must terminate with an error (or short strip), whatever they decode to
    for seed in range(20):
        bad = bytearray(raw)
        junk = np.random.default_rng(seed).integers(0, 256, len(raw) - strip_off, dtype=np.uint8).tobytes()
        bad[strip_off:] = junk
        open(tmp_path / "bad2.tif", "wb").write(bad)
        r = run("nlk-imgconv", tmp_path / "bad2.tif", tmp_path / "o.pfm")
        assert r.returncode in (0, 1), (seed, r.returncode)      # no signal, no hang
    # (3) uncompressed strip shorter than the image
    Image.fromarray(g).save(tmp_path / "u.tif")
    ur = bytearray(open(tmp_path / "u.tif", "rb").read())
    ue = entries(ur)
    ur[ue[279] + 8:ue[279] + 12] = struct.pack("<I", 40)          # StripByteCounts = 40 bytes
    open(tmp_path / "bad3.tif", "wb").write(ur)
    r = run("nlk-imgconv", tmp_path / "bad3.tif", tmp_path / "o.pfm")
    assert r.returncode != 0 and "short strip" in r.stderr
    # (4) floating-point predictor (3): decoded since round 3 (what libtiff writes when the tag is set must read back
    # exactly); on integer samples it is refused
    Image.fromarray(g).save(tmp_path / "f.tif", compression="tiff_lzw", tiffinfo={317: 3})
    r = run("nlk-imgconv", tmp_path / "f.tif", tmp_path / "o.pfm")
    fr = open(tmp_path / "f.tif", "rb").read()
    if 317 in entries(fr) and struct.unpack("<H", fr[entries(fr)[317] + 8:entries(fr)[317] + 10])[0] == 3:
        assert r.returncode == 0, r.stderr
        pf = open(tmp_path / "o.pfm", "rb").read()
        back = np.frombuffer(pf[len(pf) - g.size * 4:], "<f4").reshape(g.shape)   # (rows top to bottom, like the reference's writer)
        assert np.array_equal(back, g)
        fe = entries(fr)
        bad = bytearray(fr)
        bad[fe[339] + 8:fe[339] + 10] = struct.pack("<H", 1)          # SampleFormat = unsigned integer
        open(tmp_path / "bad4.tif", "wb").write(bad)
        r = run("nlk-imgconv", tmp_path / "bad4.tif", tmp_path / "o.pfm")
        assert r.returncode != 0 and "predictor" in r.stderr
    # (5) absurd sizes in the header
    ur = bytearray(open(tmp_path / "u.tif", "rb").read())
    ur[ue[256] + 8:ue[256] + 12] = struct.pack("<I", 0x7fffffff)
    ur[ue[257] + 8:ue[257] + 12] = struct.pack("<I", 0x7fffffff)
    open(tmp_path / "bad5.tif", "wb").write(ur)
    r = run("nlk-imgconv", tmp_path / "bad5.tif", tmp_path / "o.pfm")
    assert r.returncode != 0 and "unreasonable" in r.stderr
    open(tmp_path / "bad6.pfm", "wb").write(b"Pf\n2000000000 2000000000\n-1\n" + b"\0" * 64)
    r = run("nlk-imgconv", tmp_path / "bad6.pfm", tmp_path / "o.pfm")
    assert r.returncode != 0
    open(tmp_path / "bad7.pfm", "wb").write(b"Pf")                 # header runs into the end of the buffer
    r = run("nlk-imgconv", tmp_path / "bad7.pfm", tmp_path / "o.pfm")
    assert r.returncode != 0
