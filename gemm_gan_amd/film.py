"""FiLM-only sibling of the hot path: the reference's ``src/conditional_gan_film.py`` (F below) behind the same HIP engine
(``GG_VARIANT_FILM``, include/gemmgan.h).  Same class names and call signatures as that file:

    generator (F:97) / discriminator (F:152): forward(x, text_embedding [B,Dt], patches [B,P,Dp], padding_mask [B,P])
    WGAN_GP_model (F:207), WGAN_GP (F:226): train_disc (F:347), train_gen (F:397), train (F:434), generate_samples (F:566),
    fit (F:586; training loop, LR schedule and checkpoints only)

Differences from the cross-attention file, all inside the engine: text is ONE vector per sample (FiLM input only), the
encoder layers carry no biases (F:113-115 ``bias=False``), the conditioning vector is the encoder's CLS row (F:150), the
optimiser steps are not clipped (F:383-385 / F:428-430).  ``state_dict`` keys match the reference's.
"""
import torch

from . import _lib as L
from . import model as _m


class _FilmNet(_m._CondNet):
    _variant = "film"

    def forward(self, x, text_embedding, patches, padding_mask):
        eng = self._require_engine()
        owner = getattr(eng, "_owner", None)
        if owner is not None:
            owner._ensure_capacity(patches.shape[0], patches.shape[1], 1)
            eng = owner.engine
        dev = eng.device
        text = text_embedding.to(dev, torch.float32).reshape(patches.shape[0], 1, -1).contiguous()
        tpad = torch.zeros(patches.shape[0], 1, dtype=torch.bool, device=dev)
        with torch.no_grad():
            return eng.forward(self._engine_role, x.to(dev), patches.to(dev).contiguous(), padding_mask.to(dev), text, tpad,
                               train=self.training)


class generator(_FilmNet):
    _role = "generator"

    def __init__(self, latent_dims, embedding_dims, generator_dims, text_embedding_dims=768,
                 patches_embedding_dims=1024, negative_slope=0.0, is_bn=False):
        super().__init__(latent_dims, embedding_dims, generator_dims, text_embedding_dims,
                         patches_embedding_dims, negative_slope, is_bn)
        self.latent_dims = latent_dims
        self.generator_dims = generator_dims


class discriminator(_FilmNet):
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


class WGAN_GP(_m.WGAN_GP):
    _variant = "film"
    _clip = (0.0, 0.0)            # F:383-385, F:428-430: optimiser steps without clip_grad_norm_

    def _build_nets(self):
        return WGAN_GP_model(self.latent_dims, self.input_dims, self.embedding_dims, self.generator_dims,
                             self.discriminator_dims, self.text_embedding_dims, self.patches_embedding_dims,
                             self.negative_slope, self.is_bn)

    def _text(self, text_embedding, B):
        text = text_embedding.to(self.device, torch.float32).reshape(B, 1, -1)
        return text, self.engine.zeros(B, 1, dtype=torch.bool)

    def gradient_penalty(self, real_data, fake_data, text_embedding, patches, padding_mask):
        """F:322-345: the penalty value (0-d tensor, no autograd graph; alpha drawn by the same torch.rand call)."""
        text, tpad = self._text(text_embedding, real_data.shape[0])
        return super().gradient_penalty(real_data, fake_data, patches, padding_mask, text, tpad)

    # F:473-564 (balanced=False branch): batches are (text_embedding, gene_expression, patches, padding_mask, disease, site)
    def _generate_from_batch(self, batch):
        dev = self.device
        return self.generate_samples(batch[1].to(dev), batch[0].to(dev), batch[2].to(dev), batch[3].to(dev))

    @staticmethod
    def _labels_of(batch):
        return batch[4].detach().cpu().numpy(), batch[5].detach().cpu().numpy()

    def train_disc(self, real_data, z, text_embedding, patches, padding_mask):
        text, tpad = self._text(text_embedding, z.shape[0])
        return super().train_disc(real_data, z, text, tpad, patches, padding_mask)

    def train_gen(self, z, text_embedding, patches, padding_mask):
        text, tpad = self._text(text_embedding, z.shape[0])
        return super().train_gen(z, text, tpad, patches, padding_mask)

    def train(self, gene_expression, text_embedding, patches, padding_mask, next_batch=None):
        text, tpad = self._text(text_embedding, gene_expression.shape[0])
        if next_batch is not None:          # (gene_expression, text_embedding, patches, padding_mask) of the following call
            nt, ntp = self._text(next_batch[1], next_batch[0].shape[0])
            next_batch = (next_batch[0], nt, ntp, next_batch[2], next_batch[3])
        return super().train(gene_expression, text, tpad, patches, padding_mask, next_batch=next_batch)

    def generate_samples(self, gene_expression, text_embedding, patches, padding_mask):
        with torch.no_grad():
            self.gen.eval()
            x_real = gene_expression.clone().to(torch.float32)
            z = torch.normal(0, 1, size=(x_real.shape[0], self.latent_dims), device=self.device)
            x_gen = self.gen(z, text_embedding, patches, padding_mask)
        return x_real, x_gen

    def _fit_batch(self, data, nxt=None):
        self.train(data[1], data[0], data[2], data[3])          # F:632-637: (text_embedding, gene_expression, patches, padding_mask)
