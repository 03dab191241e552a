"""The toy LISA extrinsic-parameter likelihood used by the reference's exampleLISA regression test,
restated in numpy as an example *user plug-in* (reference: exampleLISA.cc:59-72 antenna responses,
:130-142 simpleCalculateLogLCAmpPhase).  Parameter order: d, phi, inc, lambda, beta, psi."""
import cmath
import math

FACTOR = 216147.866077
SA_INJ = complex(0.33687296665053773, 0.087978055005482114)
SE_INJ = complex(-0.12737105239204741, 0.21820079314765678)
PI = math.pi

# prior / space exactly as simple_likelihood_ni::setup (exampleLISA.cc:528-593)
TYPES = ["uni", "uni", "pol", "uni", "cpol", "uni"]
CENTERS = [1.667, PI, PI / 2, PI, 0.0, PI / 2]
SCALES = [1.333, PI, PI / 2, PI, PI / 2, PI / 2]
LIMIT, WRAP = 1, 3
BLO = [LIMIT, WRAP, LIMIT, WRAP, LIMIT, WRAP]
BHI = [LIMIT, WRAP, LIMIT, WRAP, LIMIT, WRAP]
BMIN = [0.0, 0.0, 0.0, 0.0, -PI / 2, 0.0]
BMAX = [30.0, 2 * PI, PI, 2 * PI, PI / 2, PI]


def _modes(d, phi, inc, psi, plus, cross):
    pref = 0.5 / d * math.sqrt(5 / PI)
    m22 = pref * math.cos(inc / 2) ** 4 * cmath.exp(2j * (-phi - psi)) * 0.5 * (plus + 1j * cross)
    m2m2 = pref * math.sin(inc / 2) ** 4 * cmath.exp(2j * (-phi + psi)) * 0.5 * (plus - 1j * cross)
    return m22 + m2m2


def loglike(x):
    d, phi, inc, lam, beta, psi = (float(v) for v in x)
    a_plus = 1j * (0.75 * (3 - math.cos(2 * beta)) * math.cos(2 * lam - PI / 3))
    a_cross = 1j * (3.0 * math.sin(beta) * math.sin(2 * lam - PI / 3))
    e_plus = -1j * (0.75 * (3 - math.cos(2 * beta)) * math.sin(2 * lam - PI / 3))
    e_cross = 1j * (3.0 * math.sin(beta) * math.cos(2 * lam - PI / 3))
    sa = _modes(d, phi, inc, psi, a_plus, a_cross)
    se = _modes(d, phi, inc, psi, e_plus, e_cross)
    return -0.5 * FACTOR * (abs(sa - SA_INJ) ** 2 + abs(se - SE_INJ) ** 2)
