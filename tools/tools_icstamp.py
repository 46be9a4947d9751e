import sys, ctypes as C, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, '..')
import visual_odometry_ros_amd as V
from visual_odometry_ros_amd import synthetic as S
st = S.StereoStream(); poses = st.poses(3)
L0, R0, _ = st.render_pair(poses[0]); L1, R1, _ = st.render_pair(poses[1])
ts = st.track_set(1, poses[0], poses[1])
ctx = V.Context(max_width=1241, max_height=376, max_points=4096, n_slots=4, max_level=6)
ft = V.FeatureTracker(ctx)
ctx.set_image(0, L0); ctx.set_image(1, L1)
T_cp = np.linalg.inv(ts['dT_prior'].astype(np.float64))
Xl1 = ts['Xp'] @ T_cp[:3,:3].T + T_cp[:3,3]
scale = (ts['Xp'][:,2]/Xl1[:,2]).astype(np.float32); K = st.K
prior = np.stack([K[0]*Xl1[:,0]/Xl1[:,2]+K[2], K[1]*Xl1[:,1]/Xl1[:,2]+K[3]],1).astype(np.float32)
p1, m1 = ft.trackWithPrior(0,1,ts['pts_l0'],21,6,80.0,prior)
idx = np.nonzero(m1)[0]
# run phase 1 only but with records (strict=True runs rounds too; stamps are written by phase 1 only, rounds overwrite pre1 for touched)
r = ft.trackWithScale(0,1,ts['pts_l0'][idx],scale[idx],p1[idx],None,strict_border=True)
n = len(idx); out = np.zeros((n,6),np.float32)
rc = ctx.lib.vo_debug_ic_rows(ctx.handle, out.ctypes.data_as(C.POINTER(C.c_float)), 6, n)
it = out[:,0]; ok = (it > 0) & (it <= 30) & (it == np.round(it)) & (out[:,1] > 500)
print('points', ok.sum(), 'mean iters', it[ok].mean())
per = out[ok,1:4] / it[ok,None]
t0 = out[ok,4]; t1 = out[ok,5]; base = t0.min()
print('WG start spread (us):', (t0.max()-base)/100, ' end max (us):', (t1.max()-base)/100, ' mean dur (us):', ((t1-t0)%(1<<24)).mean()/100)
d = ((t1-t0)%(1<<24))/100
for lo,hi in ((1,4),(4,8),(8,16),(16,29),(30,31)):
    sel=(it[ok]>=lo)&(it[ok]<hi)
    if sel.any(): print(f'  iters {lo}-{hi-1}: n={sel.sum()} mean dur {d[sel].mean():.1f} us, mean start {((t0[sel]-base)/100).mean():.1f} us, cyc/iter {per[sel].sum(1).mean():.0f}')
print('cycles/iter: sample %.0f  reduce+exchange %.0f  solve+err %.0f  total %.0f' % (*per.mean(0), per.sum(1).mean()))

