import sys, time
sys.path.insert(0, 'oracle'); sys.path.insert(0, 'gr-gfdm_amd/python')
import numpy as np
import gfdm_ref as R
import gfdm_amd as G
from gfdm_amd.filters import get_frequency_domain_filter
rng = np.random.default_rng(0)
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
for (M, K, L, a) in [(16,4,2,.35),(21,128,2,.35),(5,32,2,.5),(9,64,2,.2),(15,128,4,.2),(31,256,2,.1),(127,16,4,.5),(25,96,2,.35),(3,10,6,.3)]:
    taps = get_frequency_domain_filter('rrc', a, M, K, L)
    nt = R.normalize_taps(taps, M)
    B = 5
    d = ((rng.integers(0,2,(B,M*K))*2-1) + 1j*(rng.integers(0,2,(B,M*K))*2-1))/np.sqrt(2)
    mod = G.Modulator(M, K, L, taps); dem = G.Demodulator(M, K, L, taps)
    x = R.modulate(d, nt, M, K, L)
    xg = mod.modulate(d)
    feq = np.fft.fft(np.array([1,.5,.1j,.1+.05j]), M*K)[None,:]*np.exp(1j*0.01*np.arange(B))[:,None]
    xe = np.fft.ifft(np.fft.fft(x,axis=-1)*feq,axis=-1)
    y = R.demodulate(xe, nt, M,K,L, feq); yg = dem.demodulate_equalize(xe, feq)
    y0 = R.demodulate(x, nt, M,K,L); y0g = dem.demodulate(x)
    S = R.fft_filter_downsample(x, nt, M,K,L); Sg = dem.fft_filter_downsample(x)
    c = R.cancel_sc_interference(d, S, R.ic_filter_taps(nt,M,L), M, K); cg = dem.cancel_sc_interference(d, S)
    t = R.transform_subcarriers_to_td(S, M, K); tg = dem.transform_subcarriers_to_td(S)
    adv = G.AdvancedReceiver(M,K,L,taps,np.arange(K),2,R.qpsk_points(), do_phase_compensation=1)
    a2 = R.advanced_receive(xe, nt, M,K,L, np.arange(K), R.qpsk_points(), 2, f_eq=feq, kind='qpsk', do_phase_compensation=1)
    a2g = adv.demodulate_equalize(xe, feq)
    print(M,K,L,mod.kernel_name(),"mod %.2e demod %.2e demod_eq %.2e ffd %.2e td %.2e cancel %.2e adv %.2e"%(rel(xg,x), rel(y0g,y0), rel(yg,y), rel(Sg,S), rel(tg,t), rel(cg,c), rel(a2g,a2)), flush=True)
