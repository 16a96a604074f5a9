"""diff-svc hot path, MI355X-native: DDPM/PLMS sampler, DiffNet denoiser, NSF-HiFiGAN vocoder and the
STFT/mel front-end as hand-written HIP kernels for gfx950 behind a C ABI (include/dsvc.h).

Host side mirrors the reference's plugin seams:
  * ``diffsvc_amd.denoiser.DiffNetHip``         <-> network/diff/net.py  DiffNet
  * ``diffsvc_amd.sampler.GaussianDiffusionHip`` <-> network/diff/diffusion.py  GaussianDiffusion
  * ``diffsvc_amd.vocoder.NsfHifiGANHip``        <-> network/vocoders/nsf_hifigan.py  NsfHifiGAN
"""
__version__ = "0.1.0"
