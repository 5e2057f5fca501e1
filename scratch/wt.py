# wave start/finish stamps of the k_extend launches of one C3 frame (diagnostic build -DMI_WAVE_TIMES)
import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(8, 8192, 4)
for lv in range(8):
    t0, t1 = a[lv, :, 0].astype(np.int64), a[lv, :, 1].astype(np.int64)
    ok = t1 > 0
    if not ok.any(): continue
    s, e = t0[ok], t1[ok]
    hw = a[lv, ok, 2]; it = a[lv, ok, 3].astype(np.int64)
    base = s.min(); s = (s - base) / 100.0; e = (e - base) / 100.0  # us (100 MHz)
    dur = e.max()
    busy = (e - s).sum() / (ok.sum() * dur)
    qs = np.percentile(e, [0, 5, 25, 50, 75, 95, 100])
    print(f"level {lv}: waves {ok.sum()} span {dur:.1f} us  start max {s.max():.1f}  finish pct (0,5,25,50,75,95,100) {np.round(qs,1)}  wave-time used {busy:.3f}")
    print(f"   iterations per wave: min {it.min()} median {np.median(it):.0f} max {it.max()}  corr(iters, finish) {np.corrcoef(it, e)[0,1]:.3f}")
    hwid = (hw & 0xffffffff).astype(np.int64); xcc = ((hw >> 32) & 0xf).astype(np.int64)
    cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    uk, cnt = np.unique(key, return_counts=True)
    print(f"   distinct CUs {len(uk)}  waves per CU: min {cnt.min()} max {cnt.max()}  hist {np.bincount(cnt)[cnt.min():]}")
    # finish time by waves-per-CU class and by XCD
    per = dict(zip(uk, cnt)); wpc = np.array([per[k] for k in key])
    for c in np.unique(wpc): print(f"   CUs with {c} waves: mean finish {e[wpc==c].mean():.1f} us, mean us/iter {((e-s)[wpc==c]/it[wpc==c]).mean():.3f}")
    print("   by XCD mean finish:", np.round([e[xcc==x].mean() for x in range(8)], 1))
    late = s > 50
    if late.any(): print(f"   late starters: {late.sum()} waves, mean start {s[late].mean():.1f}")
