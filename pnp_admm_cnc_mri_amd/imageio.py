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


MASK_NAMES = ['Q_Random30', 'Q_Radial30', 'Q_Cartesian30']


def load_mask_mat(path, check=True):
    """One CS_MRI/Q_*.mat sampling pattern -> float64 [H,W] 0/1 mask in un-shifted FFT layout
    (what S4:185 takes as `Q1`).  The files carry up to three views of the same pattern:
        Q1     uint8 [H,W], un-shifted (DC at [0,0])       -- the one the solvers read
        Q11    uint8 [H,W] = fftshift(Q1) (DC centred)     -- for display
        OMEGA  int32 [n,1], MATLAB find(Q1): 1-based column-major indices, ascending (absent for radial)
    Q1 is used when present; a file holding only Q11 or OMEGA (+ a square size inferred from the
    largest index is NOT attempted: pass Q1/Q11) is converted.  With check=True the views that are
    present must agree, otherwise ValueError."""
    import scipy.io as sio
    d = sio.loadmat(path)
    q1, q11, om = d.get('Q1'), d.get('Q11'), d.get('OMEGA')
    if q1 is None and q11 is None:
        raise ValueError('%s holds neither Q1 nor Q11' % path)
    if q1 is None:
        q1 = np.fft.ifftshift(np.asarray(q11))
    q1 = (np.asarray(q1) != 0)
    if check:
        if q11 is not None and not np.array_equal(np.fft.fftshift(q1), np.asarray(q11) != 0):
            raise ValueError('%s: Q11 is not fftshift(Q1)' % path)
        if om is not None:
            want = np.flatnonzero(q1.T.ravel()) + 1                   # MATLAB find(): column-major, 1-based
            if not np.array_equal(np.asarray(om).ravel().astype(np.int64), want):
                raise ValueError('%s: OMEGA is not find(Q1)' % path)
    return q1.astype(np.float64)


def load_cs_mri(root='CS_MRI', check=True):
    """-> (mask [3,256,256] float64 for Q_Random30/Q_Radial30/Q_Cartesian30, noises complex128*3.0)
    exactly as S4:182-191 builds them (Q1 as float64; k-space noise scaled by 3.0)."""
    import scipy.io as sio
    mask = np.array([load_mask_mat(os.path.join(root, n + '.mat'), check) for n in MASK_NAMES])
    noises = sio.loadmat(os.path.join(root, 'noises.mat')).get('noises').astype(np.complex128) * 3.0
    return mask, noises


def save_mask_mat(path, q1, with_omega=True):
    """Write a sampling pattern in the reference's .mat layout (Q1, Q11, OMEGA)."""
    import scipy.io as sio
    q1 = (np.asarray(q1) != 0).astype(np.uint8)
    d = {'Q1': q1, 'Q11': np.fft.fftshift(q1)}
    if with_omega:
        d['OMEGA'] = (np.flatnonzero(q1.T.ravel()) + 1).astype(np.int32)[:, None]
    sio.savemat(path, d, do_compression=True)
