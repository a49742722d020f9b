"""The reference's training-data generator (csv_generator.py) with its per-batch pixel and anchor work on the device.

Same class names, constructor arguments and method names as the reference's Generator / CSVGenerator, so RetinaNet.py's train()
(:234-247) constructs it unchanged; what differs is where the work runs:

  host   CSV rows, grouping / shuffling, annotation filtering, the 3x3 augmentation matrices and box corners (a few flops)
  device per page: uint8 upload -> [rtn_warp_affine_u8, when a transform generator is given] -> rtn_resize_cubic, which fuses the
         x/127.5-1 normalisation (preprocess_image 'custom_tf'), the INTER_CUBIC resize and the write into the zero-padded batch
         canvas of compute_inputs;  per batch: rtn_anchor_targets (anchors, IoU, assignment, box deltas) -> regression, labels

__getitem__ returns (inputs, [regression, labels]) as CUDA tensors the Trainer consumes directly (output='numpy' gives the
reference's NumPy float32 arrays instead).  The generator owns its library handle and HIP stream, so `batches()` can build the
next batches on a background thread while the training step runs; the consumer's stream waits on the batch's event.

There is no CPU fallback: without the GPU library this module raises.
"""
import csv
import ctypes as C
import os
import queue
import random
import threading
import warnings

import numpy as np
import torch

try:                                    # as a module of the package ...
    from .model import Parameters, _rt
    from .model import anchors as _anchors
    from .model.transform import adjust_transform_for_image, invert_affine, transform_aabb, warp_codes
    from .model.utils import compute_resize_scale
except ImportError:                     # ... or top-level, with the package directory on sys.path like the reference's layout
    from model import Parameters, _rt
    from model import anchors as _anchors
    from model.transform import adjust_transform_for_image, invert_affine, transform_aabb, warp_codes
    from model.utils import compute_resize_scale

L = _rt.L


class ImageRecord:
    """One dataset entry (the role FasterRCNN/Shapes.py's Image + GroundTruthBox play for csv_generator.py:16-52)."""
    __slots__ = ("name", "image_path", "width", "height", "boxes", "class_names")

    def __init__(self, name, image_path, width, height, boxes, class_names):
        self.name, self.image_path, self.width, self.height = name, image_path, int(width), int(height)
        self.boxes = np.asarray(boxes, np.float64).reshape(-1, 4)
        self.class_names = list(class_names)


def read_image_bgr(path):
    """cv2.imread(path): uint8 (H,W,3) in B,G,R order.  Decoded with Pillow (OpenCV is not a dependency of this package)."""
    from PIL import Image
    with Image.open(path) as im:
        rgb = np.asarray(im.convert("RGB"))
    return np.ascontiguousarray(rgb[:, :, ::-1])


def _image_size(path):
    from PIL import Image
    with Image.open(path) as im:
        return im.size[1], im.size[0]


def _read_annotations(csv_data_file, image_dir, codeTesting=False):
    """csv_generator.py:16-52.  The first line of the CSV is skipped; rows are image_id,xmin,ymin,xmax,ymax,label; entries come
    out in sorted image_id order (the reference iterates a pandas groupby); ids without a *.png file in image_dir are dropped
    with the reference's warning."""
    present = set(n for n in os.listdir(image_dir) if n.endswith('.png'))
    per_image = {}
    with open(csv_data_file, newline='') as f:
        rows = csv.reader(f)
        next(rows, None)
        for r in rows:
            if r:
                per_image.setdefault(r[0], []).append(r)
    records = []
    for name in sorted(per_image):
        if name not in present:
            print("WARNING: Image for the entry {} not found in the data directory".format(name))
            continue
        path = os.path.join(image_dir, name)
        h, w = _image_size(path)
        rows = per_image[name]
        records.append(ImageRecord(name, path, w, h, [[float(v) for v in r[1:5]] for r in rows], [r[5] for r in rows]))
        if codeTesting:
            break
    return records


class Generator:
    """csv_generator.py:55-415."""

    def __init__(self, transform_generator=None, visual_effect_generator=None, batch_size=1, group_method='random',
                 shuffle_groups=True, image_min_side=800, image_max_side=1333, transform_parameters=None,
                 compute_anchor_targets=None, compute_shapes=None, preprocess_image=None, config=False,
                 dtype=torch.bfloat16, output='device', device=None):
        if compute_anchor_targets not in (None, _anchors.anchor_targets_bbox) or compute_shapes not in (None, _anchors.guess_shapes) \
                or preprocess_image is not None:
            raise ValueError("the device generator implements the reference's default handlers (anchor_targets_bbox, guess_shapes, "
                             "preprocess_image 'custom_tf'); custom callables are not supported")
        if output not in ('device', 'numpy'):
            raise ValueError("output must be 'device' or 'numpy'")
        if not torch.cuda.is_available():
            raise RuntimeError("csv_generator runs its batch work on a ROCm GPU through librtn.so: no CPU fallback exists")
        self.transform_generator = transform_generator
        self.visual_effect_generator = visual_effect_generator        # accepted and unused, like the reference (csv_generator.py:385)
        self.batch_size = int(batch_size)
        self.group_method = group_method
        self.shuffle_groups = shuffle_groups
        self.image_min_side = image_min_side
        self.image_max_side = image_max_side
        self.transform_parameters = transform_parameters
        self.load_parm_from_config = config
        self.dtype = dtype
        self.output = output
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self._h = L.Handle(self.device.index)
        self._stream = torch.cuda.Stream(device=self.device)
        self._h.set_stream(self._stream.cuda_stream)
        self._lock = threading.Lock()
        self.group_images()
        if self.shuffle_groups:
            self.on_epoch_end()

    # -- dataset interface (overridden by CSVGenerator) -------------------------------------------------------------------------
    def size(self):
        raise NotImplementedError('size method not implemented')

    def num_classes(self):
        raise NotImplementedError('num_classes method not implemented')

    def has_label(self, label):
        raise NotImplementedError('has_label method not implemented')

    def has_name(self, name):
        raise NotImplementedError('has_name method not implemented')

    def name_to_label(self, name):
        raise NotImplementedError('name_to_label method not implemented')

    def label_to_name(self, label):
        raise NotImplementedError('label_to_name method not implemented')

    def image_aspect_ratio(self, image_index):
        raise NotImplementedError('image_aspect_ratio method not implemented')

    def load_image(self, image_index):
        raise NotImplementedError('load_image method not implemented')

    def load_annotations(self, image_index):
        raise NotImplementedError('load_annotations method not implemented')

    # -- host side --------------------------------------------------------------------------------------------------------------
    def on_epoch_end(self):
        if self.shuffle_groups:
            random.shuffle(self.groups)

    def group_images(self):
        """csv_generator.py:159-171: order by `group_method`, cut into batches, the last one wrapping round to the start."""
        n = self.size()
        order = list(range(n))
        if self.group_method == 'random':
            random.shuffle(order)
        elif self.group_method == 'ratio':
            order.sort(key=self.image_aspect_ratio)
        bs = self.batch_size
        self.groups = [[order[j % n] for j in range(start, start + bs)] for start in range(0, n, bs)]

    def load_image_group(self, group):
        return [self.load_image(i) for i in group]

    def load_annotations_group(self, group):
        out = [self.load_annotations(i) for i in group]
        for a in out:
            assert isinstance(a, dict), '\'load_annotations\' should return a list of dictionaries, received: {}'.format(type(a))
            assert 'labels' in a and 'bboxes' in a, '\'load_annotations\' should return a list of dictionaries that contain \'labels\' and \'bboxes\'.'
        return out

    def filter_annotations(self, image_group, annotations_group, group):
        """csv_generator.py:192-218: drop boxes with x2<=x1, y2<=y1, a negative corner or a far corner past the image."""
        for i, (image, ann) in enumerate(zip(image_group, annotations_group)):
            b = ann['bboxes']
            h, w = image.shape[0], image.shape[1]
            bad = np.where((b[:, 2] <= b[:, 0]) | (b[:, 3] <= b[:, 1]) | (b[:, 0] < 0) | (b[:, 1] < 0) | (b[:, 2] > w) | (b[:, 3] > h))[0]
            if len(bad):
                warnings.warn('Image with id {} (shape {}) contains the following invalid boxes: {}.'.format(group[i], tuple(image.shape), b[bad, :]))
                for k in list(ann.keys()):
                    ann[k] = np.delete(ann[k], bad, axis=0)
        return image_group, annotations_group

    def generate_anchors(self, image_shape):
        """csv_generator.py:339-350 -> (N,4) float64 on the host (the batch path itself keeps anchors on the device)."""
        return _anchors.anchors_for_shape(image_shape, anchor_params=self._anchor_params())

    def _anchor_params(self):
        if self.load_parm_from_config:
            return _anchors.AnchorParameters(sizes=Parameters.sizes, strides=Parameters.strides, ratios=Parameters.ratios, scales=Parameters.scales)
        return _anchors.AnchorParameters_default

    # -- device side ------------------------------------------------------------------------------------------------------------
    def _upload(self, image):
        if isinstance(image, torch.Tensor):
            t = image.to(self.device)
        else:
            a = np.ascontiguousarray(image)
            if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
                raise ValueError("pages must be uint8 (H,W,3) arrays as cv2.imread returns them, got %s %s" % (a.dtype, a.shape))
            t = torch.from_numpy(a).to(self.device, non_blocking=True)
        if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
            raise ValueError("pages must be uint8 (H,W,3)")
        return t.contiguous()

    def random_transform_group_entry(self, image, annotations, transform=None):
        """csv_generator.py:248-265 on a device page: warp the pixels with rtn_warp_affine_u8, move the box corners on the host."""
        if transform is None and not self.transform_generator:
            return image, annotations
        if transform is None:
            transform = adjust_transform_for_image(next(self.transform_generator), image, self.transform_parameters.relative_translation)
        inter, border, cval = warp_codes(self.transform_parameters)
        src = self._upload(image)
        dst = torch.empty_like(src)
        inv = np.ascontiguousarray(invert_affine(transform))
        H, W, Cc = src.shape
        self._h.check(L.lib.rtn_warp_affine_u8(self._h.raw, src.data_ptr(), H, W, Cc, inv.ctypes.data_as(C.c_void_p), inter, border,
                                               cval.ctypes.data_as(C.c_void_p), dst.data_ptr()))
        boxes = annotations['bboxes'].copy()
        for i in range(boxes.shape[0]):
            boxes[i, :] = transform_aabb(transform, boxes[i, :])
        annotations['bboxes'] = boxes
        return dst, annotations

    def random_transform_group(self, image_group, annotations_group):
        assert len(image_group) == len(annotations_group)
        for i in range(len(image_group)):
            image_group[i], annotations_group[i] = self.random_transform_group_entry(image_group[i], annotations_group[i])
        return image_group, annotations_group

    def resize_image(self, image):
        """utils.resize_image (model/utils.py:140-154) of one page -> (device tensor, scale); compute_input_output uses the fused
        batch form below instead."""
        src = self._upload(image)
        scale = compute_resize_scale(src.shape, self.image_min_side, self.image_max_side)
        ho, wo = int(np.rint(src.shape[0] * scale)), int(np.rint(src.shape[1] * scale))
        out = torch.empty(ho, wo, 3, dtype=torch.float32, device=self.device)
        self._h.check(L.lib.rtn_resize_cubic(self._h.raw, src.data_ptr(), 2, src.shape[0], src.shape[1], 3, float(scale), out.data_ptr(),
                                             L.RTN_F32, ho, wo, wo * 3))
        return out, scale

    def compute_inputs(self, pages, scales=None):
        """preprocess_group (normalise + resize, csv_generator.py:289-316) fused with compute_inputs (:320-336): each uint8 device
        page lands, resized, in the top-left corner of one zero canvas of the largest resized shape.  Returns (canvas, shapes)."""
        if scales is None:
            scales = [compute_resize_scale(p.shape, self.image_min_side, self.image_max_side) for p in pages]
        shapes = [(int(np.rint(p.shape[0] * s)), int(np.rint(p.shape[1] * s))) for p, s in zip(pages, scales)]
        Hm, Wm = max(s[0] for s in shapes), max(s[1] for s in shapes)
        canvas = torch.zeros(len(pages), Hm, Wm, 3, dtype=self.dtype, device=self.device)
        code = L.RTN_BF16 if self.dtype == torch.bfloat16 else L.RTN_F32
        for i, (p, s, (ho, wo)) in enumerate(zip(pages, scales, shapes)):
            self._h.check(L.lib.rtn_resize_cubic(self._h.raw, p.data_ptr(), 2, p.shape[0], p.shape[1], 3, float(s), canvas[i].data_ptr(),
                                                 code, ho, wo, Wm * 3))
        return canvas, shapes

    def compute_targets(self, canvas_shape, image_shapes, annotations_group):
        """compute_targets (csv_generator.py:353-370) -> device (B,N,5), (B,N,K+1) float32; anchors are generated on the device."""
        ap = self._anchor_params()
        levels = [3, 4, 5, 6, 7]
        shapes = _anchors.guess_shapes(canvas_shape, levels)
        bases = [_anchors.generate_anchors(base_size=ap.sizes[i], ratios=ap.ratios, scales=ap.scales) for i in range(len(levels))]
        cfg, N = _anchors._cfg(shapes, ap.strides[:len(levels)], bases)
        B, K = len(image_shapes), self.num_classes()
        gb = np.zeros((B, L.RTN_MAX_GT, 4), np.float64)
        gl = np.zeros((B, L.RTN_MAX_GT), np.int32)
        gc = np.zeros((B,), np.int32)
        hw = np.asarray(image_shapes, np.int32).reshape(B, 2)
        for i, ann in enumerate(annotations_group):
            n = ann['bboxes'].shape[0]
            if n > L.RTN_MAX_GT:
                raise ValueError("at most %d ground-truth boxes per image are supported, got %d" % (L.RTN_MAX_GT, n))
            gb[i, :n], gl[i, :n], gc[i] = ann['bboxes'], np.asarray(ann['labels']).astype(np.int32), n
        dev = [torch.from_numpy(a).to(self.device, non_blocking=True) for a in (gb, gl, gc, hw)]
        reg = torch.empty(B, N, 5, dtype=torch.float32, device=self.device)
        lab = torch.empty(B, N, K + 1, dtype=torch.float32, device=self.device)
        self._h.check(L.lib.rtn_anchor_targets(self._h.raw, C.byref(cfg), B, K, dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(),
                                               dev[3].data_ptr(), Parameters.negative_overlap, Parameters.positive_overlap,
                                               reg.data_ptr(), lab.data_ptr()))
        self._keep = dev
        return [reg, lab]

    def compute_input_output(self, group):
        """csv_generator.py:373-398.  Everything after the file reads runs on this generator's stream."""
        image_group = self.load_image_group(group)
        annotations_group = self.load_annotations_group(group)
        image_group, annotations_group = self.filter_annotations(image_group, annotations_group, group)
        with self._lock, torch.cuda.stream(self._stream):
            self._h.set_stream(self._stream.cuda_stream)
            image_group, annotations_group = self.random_transform_group(image_group, annotations_group)
            pages = [self._upload(im) for im in image_group]
            scales = [compute_resize_scale(p.shape, self.image_min_side, self.image_max_side) for p in pages]
            for ann, s in zip(annotations_group, scales):
                ann['bboxes'] = ann['bboxes'] * s
            inputs, shapes = self.compute_inputs(pages, scales)
            targets = self.compute_targets(tuple(inputs.shape[1:]), shapes, annotations_group)
            done = torch.cuda.Event()
            done.record(self._stream)
        self.last_annotations = annotations_group
        return inputs, targets, done

    def _deliver(self, inputs, targets, done):
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(done)
        for t in [inputs] + targets:
            t.record_stream(cur)
        if self.output == 'numpy':
            torch.cuda.synchronize(self.device)
            return inputs.float().cpu().numpy(), [t.cpu().numpy() for t in targets]
        return inputs, targets

    def __len__(self):
        return len(self.groups)

    def __getitem__(self, index):
        return self._deliver(*self.compute_input_output(self.groups[index]))

    def batches(self, prefetch=2, epochs=1):
        """Iterate `epochs` passes over the groups with up to `prefetch` batches built ahead on a background thread (the role of
        Keras' OrderedEnqueuer with workers=1 in RetinaNet.py:268-278); on_epoch_end() runs between passes."""
        q = queue.Queue(maxsize=max(1, int(prefetch)))
        stop = threading.Event()

        def put(item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    pass
            return False

        def work():
            try:
                torch.cuda.set_device(self.device)
                for e in range(epochs):
                    for g in list(self.groups):
                        if not put(("ok", self.compute_input_output(g))):
                            return
                    self.on_epoch_end()
                put(("end", None))
            except BaseException as exc:        # surfaced on the consumer's thread
                put(("err", exc))

        t = threading.Thread(target=work, daemon=True)
        t.start()
        try:
            while True:
                kind, item = q.get()
                if kind == "end":
                    return
                if kind == "err":
                    raise item
                yield self._deliver(*item)
        finally:
            stop.set()
            t.join()

    def close(self):
        torch.cuda.synchronize(self.device)
        self._h.close()


class CSVGenerator(Generator):
    """csv_generator.py:418-512: a dataset described by a CSV of image_id,xmin,ymin,xmax,ymax,label rows and a directory of pages."""

    def __init__(self, csv_data_file, image_dir, class_mapping, **kwargs):
        self.image_dir = image_dir
        if self.image_dir is None:
            self.image_dir = os.path.join(os.path.dirname(csv_data_file), os.path.splitext(os.path.basename(csv_data_file))[0])
        self.classes = class_mapping
        self.labels = {v: k for k, v in self.classes.items()}
        self.image_data = _read_annotations(csv_data_file, self.image_dir)
        self.image_names = [r.name for r in self.image_data]
        super(CSVGenerator, self).__init__(**kwargs)

    def size(self):
        return len(self.image_data)

    def num_classes(self):
        return max(self.classes.values()) + 1

    def has_label(self, label):
        return label in self.labels

    def has_name(self, name):
        return name in self.classes

    def name_to_label(self, name):
        return self.classes[name]

    def label_to_name(self, label):
        return self.labels[label]

    def image_path(self, image_index):
        return self.image_data[image_index].image_path

    def image_aspect_ratio(self, image_index):
        r = self.image_data[image_index]
        return float(r.width) / float(r.height)

    def load_image(self, image_index):
        return read_image_bgr(self.image_data[image_index].image_path)

    def load_annotations(self, image_index):
        """csv_generator.py:497-512 -> {'labels': (G,), 'bboxes': (G,4)} float64."""
        r = self.image_data[image_index]
        return {'labels': np.array([self.name_to_label(n) for n in r.class_names], np.float64).reshape(-1),
                'bboxes': r.boxes.astype(np.float64).copy()}
