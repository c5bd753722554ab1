"""Loader of the reference's Taipei example (BASELINE.json configs[0]; the data files under
tests/golden/taipei/ are the reference's own example inputs, example_smoothing_clean/).

Builds the argument set of the first CalSurfG call exactly as the reference's host program does
(main.f90:134-335): fp32 colatitude / longitude in radians with pi = 3.1415926535898, period slots
Rc | Rg | Lc | Lg, sources counted per slot in file order.  Returns the same dict layout as
synth.boundary_case().
"""
import os

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "taipei")


def _vals(line):
    return line.split("c:")[0].split()


def load(directory=HERE, model="MOD"):
    f = np.float32
    with open(os.path.join(directory, "DSurfTomo.in")) as fh:
        lines = fh.read().splitlines()[3:]
    it = iter(lines)
    datafile = _vals(next(it))[0]
    nx, ny, nz = (int(v) for v in _vals(next(it))[:3])
    goxd, gozd = (f(v) for v in _vals(next(it))[:2])
    dvxd, dvzd = (f(v) for v in _vals(next(it))[:2])
    nsrc = int(_vals(next(it))[0])
    next(it)                                   # weight, damp
    minthk = f(_vals(next(it))[0])             # "sablayers"
    next(it); next(it)                         # velocity bounds, max iteration
    spfra = float(_vals(next(it))[0])
    per = []
    for _ in range(4):
        k = int(_vals(next(it))[0])
        per.append(np.array([float(v) for v in next(it).split()[:k]], np.float64) if k > 0 else np.zeros(0))
    ifsyn = int(_vals(next(it))[0])
    noiselevel = f(_vals(next(it))[0])
    kRc, kRg, kLc, kLg = (len(p) for p in per)
    kmax = kRc + kRg + kLc + kLg
    nrc = nsrc
    pi = f(3.1415926535898)
    scxf = np.zeros((nsrc, kmax), f, order="F"); sczf = np.zeros((nsrc, kmax), f, order="F")
    rcxf = np.zeros((nrc, nsrc, kmax), f, order="F"); rczf = np.zeros((nrc, nsrc, kmax), f, order="F")
    periods = np.zeros((nsrc, kmax), np.int32, order="F"); wavetype = np.zeros((nsrc, kmax), np.int32, order="F")
    igrt = np.zeros((nsrc, kmax), np.int32, order="F"); nrc1 = np.zeros((nsrc, kmax), np.int32, order="F")
    nsrc1 = np.zeros(kmax, np.int32)
    vel_obs = []
    istep = istep1 = 0
    knum = 0
    knumo = 12345
    with open(os.path.join(directory, datafile)) as fh:
        for line in fh:
            if not line.strip():
                continue
            if line[0] == "#":
                t = line[1:].split()
                lat, lon, period, wavetp, veltp = f(t[0]), f(t[1]), int(t[2]), int(t[3]), int(t[4])
                if wavetp == 2 and veltp == 0: knum = period
                if wavetp == 2 and veltp == 1: knum = kRc + period
                if wavetp == 1 and veltp == 0: knum = kRg + kRc + period
                if wavetp == 1 and veltp == 1: knum = kLc + kRg + kRc + period
                if knum != knumo:
                    istep = 0
                istep += 1
                istep1 = 0
                scxf[istep - 1, knum - 1] = (f(90.0) - lat) * pi / f(180.0)
                sczf[istep - 1, knum - 1] = lon * pi / f(180.0)
                periods[istep - 1, knum - 1] = period
                wavetype[istep - 1, knum - 1] = wavetp
                igrt[istep - 1, knum - 1] = veltp
                nsrc1[knum - 1] = istep
                knumo = knum
            else:
                t = line.split()
                lat, lon = f(t[0]), f(t[1])
                istep1 += 1
                rcxf[istep1 - 1, istep - 1, knum - 1] = (f(90.0) - lat) * pi / f(180.0)
                rczf[istep1 - 1, istep - 1, knum - 1] = lon * pi / f(180.0)
                nrc1[istep - 1, knum - 1] = istep1
                vel_obs.append(float(t[2]))
    with open(os.path.join(directory, model)) as fh:
        tok = fh.read().split()
    if model == "MOD":
        depz = np.array(tok[:nz], f)
        tok = tok[nz:]
    else:
        depz = load(directory, "MOD")["depz"]
    vels = np.asfortranarray(np.array(tok[:nx * ny * nz], f).reshape(nz, ny, nx).transpose(2, 1, 0))   # vsf(i, j, k)
    return dict(nx=nx, ny=ny, nz=nz, nparpi=(nx - 2) * (ny - 2) * (nz - 1), vels=vels, goxd=goxd, gozd=gozd, dvxd=dvxd, dvzd=dvzd,
                kRc=kRc, kRg=kRg, kLc=kLc, kLg=kLg, tRc=per[0], tRg=per[1], tLc=per[2], tLg=per[3], wavetype=wavetype, igrt=igrt,
                periods=periods, depz=depz, minthk=minthk, scxf=scxf, sczf=sczf, rcxf=rcxf, rczf=rczf, nrc1=nrc1, nsrcsurf1=nsrc1,
                kmax=kmax, nsrcsurf=nsrc, nrcf=nrc, ndata=int(nrc1.sum()), spfra=spfra, ifsyn=ifsyn, noiselevel=noiselevel,
                vel_obs=np.array(vel_obs, f))
