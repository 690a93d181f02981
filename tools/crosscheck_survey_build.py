import sys, os, subprocess, numpy as np
sys.path.insert(0, '/root/repo/oracle')
import oracle as O

def wpfm(path, a):
    a = np.ascontiguousarray(a, np.float32)
    h, w = a.shape[:2]; ch = 1 if a.ndim == 2 else a.shape[2]
    with open(path, 'wb') as f:
        f.write(b'%s\n%d %d\n-1.0\n' % (b'PF' if ch == 3 else b'Pf', w, h)); f.write(a.tobytes())
def rpfm(path):
    with open(path, 'rb') as f:
        t = f.readline().strip(); w, h = map(int, f.readline().split()); sc = float(f.readline())
        ch = 3 if t == b'PF' else 1
        a = np.frombuffer(f.read(), np.float32).reshape(h, w, ch)
    return a
rng = np.random.default_rng(0)
def clean(w, h, ch, shift=0):
    y, x = np.mgrid[0:h, 0:w].astype(np.float64); x = x + shift
    im = 128 + 60*np.sin(x/7.0)*np.cos(y/11.0) + 40*(((x//16 + y//16) % 2) - 0.5)
    im = np.clip(im, 0, 255)
    return np.repeat(im[:, :, None], ch, 2) * (np.array([1.0, 0.8, 0.6])[:ch])
for (w, h, ch) in [(64, 64, 1), (96, 64, 3)]:
    c0 = clean(w, h, ch); c1 = clean(w, h, ch, 2)
    n0 = (c0 + 20*rng.standard_normal(c0.shape)).astype(np.float32)
    n1 = (c1 + 20*rng.standard_normal(c1.shape)).astype(np.float32)
    wpfm('n0.pfm', n0); wpfm('n1.pfm', n1)
    # reference (survey shim build, serial): frame 0 spatial FLT1+FLT2
    subprocess.check_call(['/tmp/oracle/nlkalman-flt-noomp', '-i', 'n0.pfm', '-s', '20', '--flt11', 'r_f1_0.pfm', '--flt21', 'r_f2_0.pfm'])
    subprocess.check_call(['/tmp/oracle/nlkalman-flt-noomp', '-i', 'n1.pfm', '-s', '20', '--flt10', 'r_f1_0.pfm', '--flt20', 'r_f2_0.pfm', '--flt11', 'r_f1_1.pfm', '--flt21', 'r_f2_1.pfm'])
    p1 = O.default_params(20, O.FLT1); p2 = O.default_params(20, O.FLT2)
    o0 = O.rgb2opp(n0); o1 = O.rgb2opp(n1)
    f1_0 = O.filter_frame(o0, None, None, 20, p1)
    f2_0 = O.filter_frame(o0, None, f1_0, 20, p2)
    for name, mine in [('r_f1_0.pfm', f1_0), ('r_f2_0.pfm', f2_0)]:
        r = rpfm(name); m = O.opp2rgb(mine)
        print(w, h, ch, name, 'maxabs', np.abs(r - m).max(), 'rmse', np.sqrt(((r-m)**2).mean()))
    # temporal: use the reference's own outputs as prev to decouple
    d0_1 = O.rgb2opp(rpfm('r_f1_0.pfm')); d0_2 = O.rgb2opp(rpfm('r_f2_0.pfm'))
    f1_1 = O.filter_frame(o1, d0_1, None, 20, p1)
    f2_1 = O.filter_frame(o1, d0_2, f1_1, 20, p2)
    for name, mine in [('r_f1_1.pfm', f1_1), ('r_f2_1.pfm', f2_1)]:
        r = rpfm(name); m = O.opp2rgb(mine)
        print(w, h, ch, name, 'maxabs', np.abs(r - m).max(), 'rmse', np.sqrt(((r-m)**2).mean()))
