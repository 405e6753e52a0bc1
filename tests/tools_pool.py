import sys, os
sys.path[:0]=[os.getcwd(), os.path.join(os.getcwd(),"tests")]
import scenes, time
for mb in (512, 2048, 8192, 16384):
    r = scenes.hip_scene("c2",1024,1024); r.sample_pool_mb = mb
    r.render(1024); r.reset()
    t=time.time(); r.render(1024); dt=time.time()-t
    print("pool %5d MB: launches %d  frame %.1f ms  %.0f Msamples/s"%(mb, r.last_launches, dt*1e3, 1024*1024*1024/dt/1e6))
    r.close()
