"""The oracle's receiver front end against GROUND TRUTH and against a second, independent derivation (tests/second_frontend.py: float64 numpy from TS 36.211 alone).

The oracle's DSP restates an absent dependency (srsRAN), so no reference-held vector can pin it.  What CAN be measured without srsRAN:
  * its resource grid is the float64 FFT of the samples (36.211 6.12) to single-precision rounding;
  * its channel estimate is close to the TRUE channel - the same capture rendered without noise gives the true channel on every pilot - with the smoothing gain the
    configured 5-tap Gaussian filter has in theory (sum of squared taps 0.288 = 5.4 dB), the error floor its zero-padded band edges have in theory (-27 dB), and the
    collapse on ETU that a 5-tap filter over pilots 90 kHz apart must show; an independent CRS derivation (ports 0-3, normal and extended prefix) is part of that;
  * its noise, SNR and CFO figures against the transmitter's true values (the noise figure with the known bias of the residual method).
The HIP path equals the oracle bit for bit on these streams (tests/test_gpu_parity.py), so the figures hold for the product."""
import numpy as np
import pytest

import second_frontend as S
from lsn_testlib import scenario
from parity import gen_subframes, run_oracle


def _streams(snr, model=0, doppler=0.0, cfo=0.0, nprb=50, ports=2, n=4, cp=0, seed=5, preset="cfg2"):
    kw = dict(seed=seed, nof_prb=nprb, nof_ports=ports, snr_db=float(snr), cfo_hz=float(cfo), chan_model=model, doppler_hz=float(doppler), timing_offset_samples=0.0)
    if cp:
        kw["cp"] = cp
    sc = scenario(preset, **kw)
    tti0, iq, _ = gen_subframes(sc, n)
    sc0 = dict(sc, snr_db=300.0)  # the noise has a random stream of its own: scheduling and fading are those of `sc`
    _, iq0, _ = gen_subframes(sc0, n)
    _, per_sf, _ = run_oracle(sc, tti0, iq, taps=True)
    return sc, tti0, iq, iq0, per_sf


def _nmse(sc, tti0, iq, iq0, per_sf, inner_only=False):
    """-> (oracle estimate vs truth, raw least squares vs truth) in dB, on the pilot positions of every port / rx antenna"""
    nprb, cp, num, den, numls = sc["nof_prb"], sc.get("cp", 0), 0.0, 0.0, 0.0
    nsym = 14 if cp == 0 else 12
    for i in range(iq.shape[0]):
        for rx in range(sc["nof_rx"]):
            g0, g1 = S.ofdm_demod(iq0[i, rx], nprb, cp), S.ofdm_demod(iq[i, rx], nprb, cp)
            for port in range(sc["nof_ports"]):
                ce = per_sf[i]["ce"][port, rx][:nsym]
                for (l, k, h), (_, _, h1) in zip(S.ls_pilots(g0, sc["cell_id"], nprb, port, (tti0 + i) % 10, cp), S.ls_pilots(g1, sc["cell_id"], nprb, port, (tti0 + i) % 10, cp)):
                    sl = slice(2, -2) if inner_only else slice(None)
                    num += np.sum(np.abs(ce[l, k][sl] - h[sl]) ** 2)
                    numls += np.sum(np.abs(h1[sl] - h[sl]) ** 2)
                    den += np.sum(np.abs(h[sl]) ** 2)
    return 10 * np.log10(num / den), 10 * np.log10(numls / den)


@pytest.mark.parametrize("nprb,ports,cp", [(6, 1, 0), (25, 2, 0), (100, 2, 0), (50, 4, 0), (50, 2, 1)])
def test_resource_grid_is_the_float64_fft_of_the_samples(nprb, ports, cp):
    sc, tti0, iq, _, per_sf = _streams(20.0, nprb=nprb, ports=ports, cp=cp, n=2, preset="small" if nprb < 25 else "cfg2")
    nsym = 14 if cp == 0 else 12
    for i in range(2):
        for rx in range(sc["nof_rx"]):
            mine = S.ofdm_demod(iq[i, rx], nprb, cp)
            g = per_sf[i]["grid"][rx][:nsym].astype(np.complex128)
            assert np.abs(g - mine).max() <= 2e-5 * np.sqrt(np.mean(np.abs(mine) ** 2)) * np.sqrt(iq.shape[-1] / 15), (i, rx)


@pytest.mark.parametrize("ports,cp", [(1, 0), (2, 0), (4, 0), (2, 1), (4, 1)])
def test_reference_signals_of_every_port_and_prefix_sit_where_36_211_puts_them(ports, cp):
    """an independent derivation of the CRS (sequence, v, v_shift, symbols): on a noise-free flat channel the least-squares samples of each port are ONE constant,
    and the oracle's estimate is that constant away from the band edges - a wrong c_init, shift or symbol in either derivation scatters them"""
    sc, tti0, iq, iq0, per_sf = _streams(40.0, ports=ports, cp=cp, n=2)
    nsym = 14 if cp == 0 else 12
    for rx in range(sc["nof_rx"]):
        g0 = S.ofdm_demod(iq0[1, rx], sc["nof_prb"], cp)
        for port in range(ports):
            pil = S.ls_pilots(g0, sc["cell_id"], sc["nof_prb"], port, (tti0 + 1) % 10, cp)
            allh = np.concatenate([h for _, _, h in pil])
            assert np.std(allh) < 1e-3 * np.abs(np.mean(allh)), (rx, port, np.std(allh), np.mean(allh))
            ce = per_sf[1]["ce"][port, rx][:nsym, 24:-24]
            assert np.abs(ce - np.mean(allh)).max() < 0.05 * np.abs(np.mean(allh)), (rx, port)


def test_channel_estimate_has_the_smoothing_gain_and_edge_floor_of_its_filter():
    flat10 = _nmse(*_streams(10.0, n=6))
    assert flat10[0] <= flat10[1] - 4.5, flat10                     # theory 5.4 dB (sum of squared taps 0.288), a little lost at the band edges
    flat20_inner = _nmse(*_streams(20.0, n=6), inner_only=True)
    assert flat20_inner[0] <= flat20_inner[1] - 5.0, flat20_inner
    flat20 = _nmse(*_streams(20.0, n=6))
    assert -24.0 <= flat20[0] <= -22.0, flat20                      # the zero-padded convolution attenuates the two outermost pilots of each side: a floor near -27 dB
    flat40 = _nmse(*_streams(40.0, n=4))
    assert -28.5 <= flat40[0] <= -26.0, flat40                      # ... which is all that is left without noise (theory: (0.30^2 + 0.054^2) * 2 / 100 pilots = -27.3 dB)


def test_channel_estimate_on_the_fading_profiles_of_ts_36_101():
    epa = _nmse(*_streams(20.0, model=1, doppler=5.0, n=6))
    assert epa[0] <= epa[1] - 2.5, epa                               # 0.4 us of delay spread: most of the smoothing gain survives
    eva = _nmse(*_streams(20.0, model=2, doppler=70.0, n=6))
    assert eva[0] <= eva[1] + 0.5, eva                               # 2.5 us: the filter's bias eats its gain - no better than raw least squares at 20 dB
    etu = _nmse(*_streams(20.0, model=3, doppler=70.0, n=6))
    assert etu[0] >= etu[1] + 3.0 and etu[0] >= -17.0, etu           # 5 us turns by 2.8 rad between neighbouring pilots: smoothed away, at any SNR (the reference's settings)


@pytest.mark.parametrize("cfo", [300.0, -450.0, 0.0])
def test_cfo_noise_and_snr_figures_against_the_transmitter(cfo):
    sc, tti0, iq, iq0, per_sf = _streams(20.0, cfo=cfo, n=6)
    ch = np.array([p["chest"] for p in per_sf])  # ..., noise_avg, rsrp_avg, snr_db, cfo_hz, chan_ref
    assert abs(float(np.mean(ch[:, -2])) - cfo) < 5.0, ch[:, -2]
    # true per-RE figures: noise variance 10^(-snr/10); channel power from the noise-free pilots
    hp = []
    for rx in range(sc["nof_rx"]):
        g0 = S.ofdm_demod(iq0[0, rx], sc["nof_prb"])
        for port in range(sc["nof_ports"]):
            hp += [np.mean(np.abs(h) ** 2) for _, _, h in S.ls_pilots(g0, sc["cell_id"], sc["nof_prb"], port, tti0 % 10)]
    true_snr = 10 * np.log10(np.mean(hp) / 10 ** (-2.0))
    # the residual method (|smoothed - raw|^2, no correction for a 5-tap filter) reads (1 - w0)^2 + sum of the other squared taps = 0.48 of the variance
    ratio = float(np.mean(ch[:, -5])) / 10 ** (-2.0)
    assert 0.45 <= ratio <= 0.75, ratio
    # ... so the SNR reads up to 3 dB high; a frequency offset turns the pilots over the subframe and takes some of the coherent RSRP back
    # (RSRP = |mean of the raw samples|^2: 450 Hz is 2.2 rad from the first to the last pilot symbol = -1.8 dB)
    d = float(np.mean(ch[:, -3])) - true_snr
    assert (1.0 <= d <= 3.5) if cfo == 0.0 else (-2.5 <= d <= 3.5), (ch[:, -3], true_snr)
