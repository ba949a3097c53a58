import sys; sys.path.insert(0,'.')
from raytracer_amd import abi
W,H=1920,1080
with abi.Context(0) as ctx:
    ctx.set_scene(abi.default_scene()); ctx.set_camera(abi.default_camera(W,H)); ctx.set_frame(W,H)
    p=abi.make_params(W,H,spp=64,bounces=5,mode=1)
    for i in range(5):
        img,st=ctx.render(p)
        print("kernel_ms=%.3f gather_ms=%.3f total_ms=%.3f"%(st.kernel_ms,st.gather_ms,st.total_ms))
