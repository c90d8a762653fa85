"""Image preprocessing.  CPU: the numpy restatement of Pillow's 8-bit resampler is bit-exact against Pillow, the
full pipeline against `CLIPImageProcessor.preprocess` and (reference mounted) against `process_images(..., 'pad')`; the product's
vectorised coefficient tables equal the oracle's scalar ones.  GPU: the HIP kernels equal the oracle bit for bit."""
import numpy as np
import pytest
import torch

from ml_fastvlm_amd import preprocess as PP
from oracle import preprocess_oracle as O
from oracle import ref_import

SIZES = [(100, 150, 64), (150, 100, 64), (64, 64, 64), (37, 53, 128), (480, 640, 256), (333, 1000, 96), (70, 70, 256)]


def _img(h, w, seed):
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)


@pytest.mark.parametrize("h,w,oh,ow", [(100, 150, 64, 96), (37, 53, 128, 130), (300, 200, 64, 64), (64, 64, 64, 64), (500, 333, 224, 149), (20, 1000, 64, 320)])
def test_oracle_resize_is_pillow_bit_for_bit(h, w, oh, ow):
    from PIL import Image
    a = _img(h, w, h + w)
    want = np.asarray(Image.fromarray(a).resize((ow, oh), resample=Image.BICUBIC))
    assert np.array_equal(O.resize_bicubic_u8(a, oh, ow), want)


@pytest.mark.parametrize("h,w,r", SIZES)
def test_oracle_pipeline_is_the_hf_processor(h, w, r):
    from PIL import Image
    from transformers import CLIPImageProcessor
    ip = CLIPImageProcessor(crop_size={"height": r, "width": r}, image_mean=[0.0, 0.0, 0.0], image_std=[1.0, 1.0, 1.0], size={"shortest_edge": r})
    a = _img(h, w, 3 * h + w)
    want = ip.preprocess(Image.fromarray(a), return_tensors="pt")["pixel_values"][0].numpy()
    got = O.preprocess(a, r, pad=False)
    assert got.dtype == np.float32 and np.array_equal(got, want)


@pytest.mark.skipif(not ref_import.reference_available(), reason="reference tree not mounted")
@pytest.mark.parametrize("h,w,r", SIZES[:5])
def test_oracle_pad_mode_is_the_reference_process_images(h, w, r):
    from types import SimpleNamespace
    from PIL import Image
    from transformers import CLIPImageProcessor
    ref_import.import_reference()
    from llava.mm_utils import process_images
    ip = CLIPImageProcessor(crop_size={"height": r, "width": r}, image_mean=[0.0, 0.0, 0.0], image_std=[1.0, 1.0, 1.0], size={"shortest_edge": r})
    a = _img(h, w, 5 * h + w)
    want = process_images([Image.fromarray(a)], ip, SimpleNamespace(image_aspect_ratio="pad"))[0].numpy()
    assert np.array_equal(O.preprocess(a, r, pad=True), want)


@pytest.mark.parametrize("n_in,n_out", [(150, 96), (53, 130), (1000, 320), (64, 64), (4032, 1024), (7, 64)])
def test_vectorised_coefficients_equal_the_scalar_restatement(n_in, n_out):
    b, k = PP._coeffs(n_in, n_out)
    if n_in == n_out:
        assert k.shape == (n_out, 1) and (k == 1 << 22).all() and (b[:, 0] == np.arange(n_out)).all()
        return
    ob, ok, _ = O.precompute_coeffs(n_in, n_out)
    assert np.array_equal(b, ob) and np.array_equal(k, ok)


def test_no_cpu_path():
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        PP.preprocess_image(torch.zeros(8, 8, 3, dtype=torch.uint8), 64)


@pytest.mark.gpu
@pytest.mark.parametrize("pad", [True, False])
@pytest.mark.parametrize("h,w,r", SIZES + [(1536, 2048, 1024)])
def test_hip_preprocess_equals_the_oracle_bit_for_bit(h, w, r, pad):
    a = _img(h, w, 7 * h + w + pad)
    want = O.preprocess(a, r, pad=pad)
    got = PP.preprocess_image(torch.from_numpy(a).cuda(), r, pad=pad)
    torch.cuda.synchronize()
    assert got.dtype == torch.float32 and np.array_equal(got.cpu().numpy(), want)
    half = PP.preprocess_image(torch.from_numpy(a).cuda(), r, pad=pad, dtype=torch.bfloat16)
    assert torch.equal(half.cpu(), torch.from_numpy(want).to(torch.bfloat16))


@pytest.mark.gpu
def test_hip_process_images_batch_and_strided_source():
    imgs = [_img(90, 120, 1), _img(200, 130, 2), _img(64, 64, 3)]
    want = np.stack([O.preprocess(a, 64, pad=True) for a in imgs])
    big = torch.zeros(200, 140, 3, dtype=torch.uint8).cuda()
    big[:, :130] = torch.from_numpy(imgs[1]).cuda()
    dev = [torch.from_numpy(imgs[0]).cuda(), big[:, :130], torch.from_numpy(imgs[2]).cuda()]          # the second one is a view with a row pitch
    got = PP.process_images(dev, 64, "pad")
    assert got.shape == (3, 3, 64, 64) and np.array_equal(got.cpu().numpy(), want)
    with pytest.raises(ValueError, match="grid_pinpoints"):
        PP.process_images(dev, 64, "anyres")


GRIDS = [[64, 128], [128, 64], [128, 128], [192, 64], [64, 192]]            # (width, height), multiples of the 64-px patch


@pytest.mark.skipif(not ref_import.reference_available(), reason="reference tree not mounted")
@pytest.mark.parametrize("h,w", [(100, 150), (150, 100), (64, 64), (37, 153), (300, 90), (128, 128)])
def test_oracle_anyres_is_the_reference(h, w):
    from PIL import Image
    from transformers import CLIPImageProcessor
    ref_import.import_reference()
    from llava.mm_utils import process_anyres_image, select_best_resolution
    ip = CLIPImageProcessor(crop_size={"height": 64, "width": 64}, image_mean=[0.0, 0.0, 0.0], image_std=[1.0, 1.0, 1.0], size={"shortest_edge": 64})
    a = _img(h, w, 11 * h + w)
    want = process_anyres_image(Image.fromarray(a), ip, str(GRIDS)).numpy()
    got = O.preprocess_anyres(a, 64, GRIDS)
    assert got.shape == want.shape and np.array_equal(got, want)
    assert PP._best_resolution(w, h, [tuple(g) for g in GRIDS]) == tuple(select_best_resolution((w, h), [tuple(g) for g in GRIDS]))


@pytest.mark.gpu
@pytest.mark.parametrize("h,w", [(100, 150), (150, 100), (64, 64), (37, 153), (300, 90), (500, 700)])
def test_hip_anyres_equals_the_oracle_bit_for_bit(h, w):
    a = _img(h, w, 13 * h + w)
    want = O.preprocess_anyres(a, 64, GRIDS)
    got = PP.process_anyres_image(torch.from_numpy(a).cuda(), 64, str(GRIDS))
    assert got.shape == want.shape and np.array_equal(got.cpu().numpy(), want)
    both = PP.process_images([torch.from_numpy(a).cuda()] * 2, 64, "anyres", grid_pinpoints=GRIDS)
    assert both.shape == (2,) + want.shape
