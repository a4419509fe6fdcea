"""Training step of DC-VIC on HIP kernels: a reverse-mode tape (autograd.py), the backward / loss /
optimizer kernels (csrc/train.hip via kernels.py), differentiable forwards of the trainable sub-networks and the PatchGAN
discriminator (nets.py) and the stage-3 GAN trainer with data-parallel gradient averaging (trainer.py)."""
from .autograd import Ctx, ParamGroup, Var  # noqa: F401
from .nets import DualBetaCondTamingNLayerDiscriminator  # noqa: F401
from .trainer import Adam, DualBetaCondGanDistortionVqCodeTrainer, MultiStepLR, allreduce_mean_  # noqa: F401
