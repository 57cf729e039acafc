import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import hmmsort_amd as H
K, N, T = 30, 3, 20_000
temps = np.asfortranarray(np.stack([H.create_spike_template(K, 3.0, 0.8, 0.2),
                                    H.create_spike_template(K, 4.0, 0.3, 0.2),
                                    H.create_spike_template(K, 2.5, 0.6, 0.25)], 1))
pp = [0.004, 0.002, 0.003]
y = H.create_signal(T, 0.3, pp, temps, seed=5)
for dead in (None, 1):
    lp = np.log(pp)
    if dead is not None:
        lp[dead] = -np.inf
    sm = H.StateMatrix.create(N, K, lp, False)
    H.set_option("engine", H.ENGINE_WAVE)
    plan = H.Plan(T, sm, temps, 0.3)
    dy = torch.from_numpy(y).cuda()
    dx = torch.zeros(T, dtype=torch.int16, device="cuda")
    dll = torch.zeros(1, dtype=torch.float64, device="cuda")
    plan.viterbi(dy, dx, dll)
    print("dead", dead, plan.info(), plan.diagnostics())
    import ctypes as C
    rec = (C.c_double * 64)()
    H._lib.lib().hmmsort_plan_debug_record.argtypes = [C.c_void_p, C.c_void_p]
    H._lib.lib().hmmsort_plan_debug_record(plan._h, rec)
    for i in range(3):
        r = list(rec[16 + 8 * i:24 + 8 * i])
        L = K - 1
        def name(ix):
            ix = int(ix)
            return "D0" if ix == 0 else ("ring %d j=%d" % ((ix - 1) // L, (ix - 1) % L + 1) if ix > 0 else "-")
        print("   boundary", r[0], "lo", r[1], name(r[2]), "hi", r[3], name(r[4]), "bad", name(r[5]), "pre/end at lo", r[6], r[7])
    SR = 1 + N * (K - 1)
    fn = H._lib.lib().hmmsort_plan_debug_array
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
    ve = np.zeros(5 * SR); fn(plan._h, 5, ve.ctypes.data, 5 * SR)
    vp = np.zeros(5 * SR); fn(plan._h, 6, vp.ctypes.data, 5 * SR)
    for c in range(4):
        print('   vend', c, ve[c*SR:c*SR+4], 'ring1', ve[c*SR+1+29:c*SR+1+32], 'ring2', ve[c*SR+1+58+12:c*SR+1+58+15])
        print('   vpre', c+1, vp[(c+1)*SR:(c+1)*SR+4], 'ring2', vp[(c+1)*SR+1+58+12:(c+1)*SR+1+58+15])
    plan.close()
