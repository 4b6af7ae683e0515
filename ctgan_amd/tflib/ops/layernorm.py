"""tflib.ops.layernorm - signature of TF/tflib/ops/layernorm.py:6-20.

Only config[4] (128x128 critic, LS/wgan_LSUN_Bedrooms128.py:70-72) uses it; its kernels (incl. the
second derivative the gradient penalty needs) are a later SURVEY 8 row - not built yet.
"""


def Layernorm(name, norm_axes, inputs):
    raise NotImplementedError('Layernorm (config[4] critic) is not built yet: SURVEY.md section 7.1 step 10')
