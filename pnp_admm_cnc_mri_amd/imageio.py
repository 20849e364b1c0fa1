"""Image / mask / noise I/O at the edges of the solvers.

Restates what the reference does around the hot loop, without cv2:
  * gray decode  = cv2.imread(path, 0): OpenCV's fixed-point BGR->gray on the decoded RGB
    (utils/utils_image.py:145-157);
  * re-quantisation uint2single(single2uint(uint2single(.))) (S4:91-94, utils_image.py:181-186);
  * CS_MRI/*.mat loading (S4:182-191): `Q1` masks, `noises` * 3.0;
  * modcrop(img, 8) (utils_image.py:495-508).
"""
import os

import numpy as np

IMG_EXTENSIONS = ['.jpg', '.JPG', '.jpeg', '.JPEG', '.png', '.PNG', '.ppm', '.PPM', '.bmp', '.BMP', '.tif']


def get_image_paths(dataroot):
    """Sorted recursive listing, utils/utils_image.py:66-82 (asserts like the reference)."""
    assert os.path.isdir(dataroot), '{:s} is not a valid directory'.format(dataroot)
    images = []
    for dirpath, _, fnames in sorted(os.walk(dataroot)):
        for fname in sorted(fnames):
            if any(fname.endswith(e) for e in IMG_EXTENSIONS):
                images.append(os.path.join(dirpath, fname))
    assert images, '{:s} has no valid image file'.format(dataroot)
    return sorted(images)


def imread_gray(path):
    """uint8 [H,W]; equals cv2.imread(path, cv2.IMREAD_GRAYSCALE) for 8-bit L/RGB/RGBA PNGs."""
    from PIL import Image
    im = Image.open(path)
    if im.mode in ('L', 'P', '1'):
        return np.asarray(im.convert('L'))
    rgb = np.asarray(im.convert('RGB')).astype(np.int64)
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    return ((4899 * r + 9617 * g + 1868 * b + 8192) >> 14).astype(np.uint8)


def imsave_gray(img, path):
    from PIL import Image
    a = np.asarray(img)
    if a.dtype != np.uint8:
        a = np.clip(np.rint(a), 0, 255).astype(np.uint8)
    Image.fromarray(a).save(path)


def modcrop(img, scale):
    H, W = img.shape[:2]
    return img[:H - H % scale, :W - W % scale]


def uint2single(img):
    return np.float32(img / 255.)


def single2uint(img):
    return np.uint8((img.clip(0, 1) * 255.).round())


def requantise(img_uint8):
    """uint8 -> the float32 image the solver sees (S4:91-94)."""
    return uint2single(single2uint(uint2single(np.asarray(img_uint8))))


def load_cs_mri(root='CS_MRI'):
    """-> (mask [3,256,256] float64 for Q_Random30/Q_Radial30/Q_Cartesian30, noises complex128*3.0)
    exactly as S4:182-191 builds them."""
    import scipy.io as sio
    names = ['Q_Random30', 'Q_Radial30', 'Q_Cartesian30']
    mask = np.array([sio.loadmat(os.path.join(root, n + '.mat')).get('Q1').astype(np.float64) for n in names])
    noises = sio.loadmat(os.path.join(root, 'noises.mat')).get('noises').astype(np.complex128) * 3.0
    return mask, noises
