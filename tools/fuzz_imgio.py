"""Mutation fuzzing of the image readers (host/imgio.c) under AddressSanitizer + UBSan, on the CPU:
    gcc -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -std=gnu99 -Ibwd-nlkalman_amd/host \
        -o /tmp/fz/imgconv_asan bwd-nlkalman_amd/host/main_imgconv.c bwd-nlkalman_amd/host/imgio.c -lm -ldl
    python tools/fuzz_imgio.py 15000 [seed]
Seeds: tiled / BigTIFF / deflate / planar / predictor-3 TIFFs from the test helpers, LZW / PackBits / deflate TIFFs and
PNGs written by PIL, PFM, FLO; 1-8 byte-level mutations each (biased to the headers). A run counts as bad when the
sanitizers report, the process dies on a signal or exits with anything but 0 / 1. Round 3: 18 000 runs, 0 bad."""
import os, subprocess, sys, random
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import importlib
trt = importlib.import_module('test_reference_tools')
rng = np.random.default_rng(0)
d = '/tmp/fz/seeds'
os.makedirs(d, exist_ok=True)
conv = os.path.join(ROOT, 'bwd-nlkalman_amd', 'bin', 'nlk-imgconv')
with open(d + '/a.pfm', 'wb') as f:
    f.write(b'PF\n17 13\n-1.0\n'); f.write(np.random.default_rng(0).uniform(0, 255, (13, 17, 3)).astype(np.float32).tobytes())
for ext in ('tif', 'png'):
    subprocess.run([conv, d + '/a.pfm', d + '/b.' + ext])
a = (rng.uniform(0, 255, (13, 17, 3))).astype(np.float32)
for comp in ('raw', 'deflate'):
    for planar in (False, True):
        for fp in (False, True):
            for big in (False, True):
                trt._tiled_tiff(d + f'/t_{comp}_{int(planar)}_{int(fp)}_{int(big)}.tif', a, 16, 16, comp=comp, planar=planar, big=big, fp_predictor=fp)
# a PIL-written LZW / packbits tiff with predictor 2 if PIL is there
try:
    from PIL import Image
    im = Image.fromarray((a[..., 0]).astype(np.uint8))
    im.save(d + '/p_lzw.tif', compression='tiff_lzw'); im.save(d + '/p_pb.tif', compression='packbits'); im.save(d + '/p_def.tif', compression='tiff_adobe_deflate')
    Image.fromarray(a.astype(np.uint8)).save(d + '/p_rgb.png')
    Image.fromarray((a[..., 0] * 200).astype(np.uint16)).save(d + '/p_16.png')
except Exception as e:
    print('no PIL', e)
flo = np.zeros((5, 7, 2), np.float32)
with open(d + '/f.flo', 'wb') as f:
    f.write(b'PIEH'); f.write(np.array([7, 5], np.int32).tobytes()); f.write(flo.tobytes())
seeds = sorted(os.listdir(d))
print(len(seeds), 'seeds')
exe = '/tmp/fz/imgconv_asan'
random.seed(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
n = 0
env = dict(os.environ, ASAN_OPTIONS='detect_leaks=0:abort_on_error=0')
for it in range(int(sys.argv[1])):
    s = random.choice(seeds)
    b = bytearray(open(os.path.join(d, s), 'rb').read())
    k = random.choice([1, 1, 2, 4, 8])
    for _ in range(k):
        mode = random.random()
        pos = random.randrange(min(len(b), 400)) if random.random() < 0.7 else random.randrange(len(b))
        if mode < 0.5: b[pos] = random.randrange(256)
        elif mode < 0.7: b[pos] = random.choice([0, 0xff, 0x7f, 0x80])
        elif mode < 0.85 and len(b) > 16: del b[pos:pos + random.randrange(1, 16)]
        else: b[pos:pos] = bytes(random.randrange(256) for _ in range(random.randrange(1, 8)))
    ext = os.path.splitext(s)[1]
    fn = '/tmp/fz/case' + ext
    open(fn, 'wb').write(b)
    r = subprocess.run([exe, fn, '/tmp/fz/out.pfm'], capture_output=True, timeout=20, env=env)
    n += 1
    if b'AddressSanitizer' in r.stderr or b'runtime error' in r.stderr or r.returncode < 0 or r.returncode > 1:
        bad += 1
        keep = f'/tmp/fz/crash_{bad}{ext}'
        open(keep, 'wb').write(b)
        print('CRASH', s, r.returncode, r.stderr.decode(errors='replace')[:600])
        if bad >= 5: break
print('runs', n, 'bad', bad)
