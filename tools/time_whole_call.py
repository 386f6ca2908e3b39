import sys, time, logging
sys.path.insert(0, '/root/repo')
from telescope_amd import _lib, synthetic
from telescope_amd.likelihood import TelescopeLikelihood
class O: em_epsilon=0.0; max_iter=20; pi_prior=0; theta_prior=200000
eng=_lib.Engine(0); eng.set_option('value_format',1); eng.set_option('kernel_timing',0)
eng.generate(0,50_000_000,30000,synthetic.poisson_cdf_u32(40),42,1,0.0)
tl=TelescopeLikelihood.from_engine(eng,O())
def T(f,name):
    eng.synchronize(); t=time.perf_counter(); r=f(); eng.synchronize(); print('%-40s %.2f ms'%(name,(time.perf_counter()-t)*1e3)); return r
T(lambda: tl.em(loglev=logging.DEBUG, final_lnl=False),'em 20 no lnl (first)')
T(lambda: tl.em(loglev=logging.DEBUG, final_lnl=False),'em 20 no lnl')
T(lambda: eng.final_lnl(),'final_lnl first')
T(lambda: eng.final_lnl(),'final_lnl')
T(lambda: tl.em(loglev=logging.DEBUG),'em 20 + lnl')
T(lambda: tl.em(loglev=logging.DEBUG),'em 20 + lnl again')
T(lambda: eng.em_chunk(20,0.0,False),'em_chunk 20')
T(lambda: eng.get_params(),'get_params')
