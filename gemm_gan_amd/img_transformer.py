"""Image-transformer sibling of the hot path: the reference's ``src/conditional_gan_img_transformer.py`` (I below) behind the
same HIP engine (``GG_VARIANT_IMG``, include/gemmgan.h).  Same class names and 4-argument call signatures as that file
(they equal conditional_gan_film.py's): generator (I:95) / discriminator (I:139) ``forward(x, text_embedding, patches,
padding_mask)``, WGAN_GP_model (I:192), WGAN_GP (I:211) with train_disc (I:330), train_gen (I:378), train (I:415),
generate_samples (I:549), fit (I:569; training loop, LR schedule, checkpoints).

Inside the engine: no FiLM (the text embedding is accepted and ignored, as upstream), the patch encoder is
Linear -> ReLU -> LayerNorm (I:106-110), bias-free encoder layers, CLS-row conditioning, no gradient clipping.  The
reference's init_train (I:277-286) knows 'rms_prop' and 'adam'; 'adamw' is accepted here as well.
"""
from . import film as _f


class _ImgNet(_f._FilmNet):
    _variant = "img"


class generator(_ImgNet):
    _role = "generator"

    def __init__(self, latent_dims, embedding_dims, generator_dims, text_embedding_dims=768,
                 patches_embedding_dims=1024, negative_slope=0.0, is_bn=False):
        super().__init__(latent_dims, embedding_dims, generator_dims, text_embedding_dims,
                         patches_embedding_dims, negative_slope, is_bn)
        self.latent_dims = latent_dims
        self.generator_dims = generator_dims


class discriminator(_ImgNet):
    _role = "discriminator"

    def __init__(self, vector_dims, embedding_dims, discriminator_dims, text_embedding_dims=768,
                 patches_embedding_dims=1024, negative_slope=0.0, is_bn=False):
        super().__init__(vector_dims, embedding_dims, discriminator_dims, text_embedding_dims,
                         patches_embedding_dims, negative_slope, is_bn)
        self.vector_dims = vector_dims
        self.discriminator_dims = discriminator_dims


def WGAN_GP_model(latent_dims, vector_dims, embedding_dims, generator_dims, discriminator_dims,
                  text_embedding_dims=768, patches_embedding_dims=1024, negative_slope=0.0, is_bn=False):
    gen = generator(latent_dims, embedding_dims, generator_dims, text_embedding_dims,
                    patches_embedding_dims, negative_slope, is_bn)
    disc = discriminator(vector_dims, embedding_dims, discriminator_dims, text_embedding_dims,
                         patches_embedding_dims, negative_slope, is_bn)
    return gen, disc


class WGAN_GP(_f.WGAN_GP):
    _variant = "img"

    def _build_nets(self):
        return WGAN_GP_model(self.latent_dims, self.input_dims, self.embedding_dims, self.generator_dims,
                             self.discriminator_dims, self.text_embedding_dims, self.patches_embedding_dims,
                             self.negative_slope, self.is_bn)
