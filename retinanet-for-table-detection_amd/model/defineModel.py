"""model/defineModel.py of the reference: Backbone / ResNetBackbone, the submodel and pyramid builders, retinanet,
retinanet_bbox and the resnet*_retinanet constructors (model/defineModel.py:9-402) — returning a Model object with the Keras
methods RetinaNet.py uses (compile, fit_generator, train_on_batch, predict_on_batch, load_weights, save, get_layer, ...),
executed by the HIP engine.  Weight files are .npz with the reference checkpoint's layer names ('<layer>/kernel' HWIO ...);
Keras HDF5 import needs h5py, which this image lacks."""
import os

import numpy as np
import torch

from . import _rt, initializers, layers, losses, utils
from .anchors import AnchorParameters_default

Wt = _rt.weights


class _LayerRef:
    def __init__(self, name, kind):
        self.name, self.kind, self.trainable = name, kind, True
        self.output = "%s:output" % name


class Adam:
    """ keras.optimizers.Adam(lr, clipnorm) stand-in accepted by Model.compile (RetinaNet.py:130).  `clipnorm` clips by the GLOBAL
    gradient norm (standalone Keras 2.x, the reference's `import keras`); global_clipnorm=False selects the per-tensor
    tf.clip_by_norm of tf.keras / Keras >= 2.4."""

    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, clipnorm=None, global_clipnorm=True, **kwargs):
        self.lr, self.beta_1, self.beta_2, self.epsilon, self.clipnorm = lr, beta_1, beta_2, epsilon, clipnorm
        self.global_clipnorm = bool(global_clipnorm)


class History:
    def __init__(self):
        self.history = {}


class Model:
    """The training model (outputs [regression, classification]) or, with bbox=True, the inference model
    (outputs [boxes, scores, labels])."""

    def __init__(self, backbone, num_classes, num_anchors, name='retinanet', bbox=False, base=None, anchor_params=None,
                 dtype=None, seed=0):
        self.backbone, self.num_classes, self.num_anchors, self.name = backbone, num_classes, num_anchors, name
        self.bbox, self.base = bbox, base
        self.anchor_params = anchor_params
        self.dtype = dtype or os.environ.get("RTN_DTYPE", "bf16")
        self.inputs = ["input_1"]
        self.output_names = ['filtered_detections'] * 3 if bbox else ['regression', 'classification']
        self.outputs = ["boxes", "scores", "labels"] if bbox else ["regression", "classification"]
        self.layers = [_LayerRef("input_1", "input")] + [_LayerRef(n, "conv") for (n, *_r) in Wt.conv_layers(backbone, num_classes, num_anchors)]
        self.layers += [_LayerRef(n, "merge") for n in ("regression", "classification")]
        self.stop_training = False
        self._state = base._state if base is not None else Wt.init_state(backbone, num_classes, num_anchors, seed=seed)
        self._engine = None
        self._trainer = None
        self._compiled = None

    # ---- Keras surface
    def get_layer(self, name):
        for l in self.layers:
            if l.name == name:
                return l
        if name in ("P3", "P4", "P5", "P6", "P7"):
            return _LayerRef(name, "conv")
        raise ValueError('No such layer: ' + name)

    def _root(self):
        return self.base._root() if self.base is not None else self

    def engine(self):
        root = self._root()
        if root._engine is None:
            ap = None
            if self.anchor_params is not None:
                ap = _rt.engine.AnchorParams(self.anchor_params.sizes, self.anchor_params.strides, self.anchor_params.ratios,
                                             self.anchor_params.scales)
            root._engine = _rt.engine.Engine(root.backbone, root.num_classes, root.num_anchors, dtype=root.dtype, anchor_params=ap)
            root._engine.load_state(root._state)
        return root._engine

    def compile(self, loss=None, optimizer=None, **kwargs):
        if loss is not None and set(loss) != {'regression', 'classification'}:
            raise ValueError("loss must name the 'regression' and 'classification' outputs")
        self._compiled = optimizer if optimizer is not None else Adam(lr=1e-4, clipnorm=0.001)

    def _get_trainer(self):
        root = self._root()
        if root._trainer is None:
            opt = self._compiled or Adam(lr=1e-4, clipnorm=0.001)
            pg = None
            if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
                pg = torch.distributed.group.WORLD
            root._trainer = _rt.trainer.Trainer(self.engine(), lr=opt.lr, clipnorm=(opt.clipnorm or 0.0), beta1=opt.beta_1,
                                                beta2=opt.beta_2, eps=opt.epsilon, process_group=pg,
                                                global_clip=getattr(opt, "global_clipnorm", True))
            tr, eng = root._trainer, root._engine
            for l in root.layers:                                 # utils.freeze / freeze_model: non-trainable layers get no update
                if l.kind == "conv" and not l.trainable:
                    lo = eng.layout[l.name]
                    tr.gscale[lo["woff"]:lo["woff"] + lo["rows"] * lo["K"]] = 0
                    tr.gscale[tr.NW + lo["boff"]:tr.NW + lo["boff"] + lo["rows"]] = 0
        return root._trainer

    def predict_on_batch(self, x):
        x = _rt.dev(x, torch.float32) if not (isinstance(x, np.ndarray) and x.dtype == np.uint8) else _rt.dev(x, torch.uint8)
        eng = self.engine()
        if self.bbox:
            b, s, l = eng.detect(x)
            return [_rt.host(b).copy(), _rt.host(s).copy(), _rt.host(l).copy()]
        r, c = eng.forward(x)
        return [_rt.host(r).copy(), _rt.host(c).copy()]

    predict = predict_on_batch

    def predict_generator(self, generator, steps=None, in_flight=2, **kwargs):
        """keras.Model.predict_generator for the inference model (retinanet_bbox, model/defineModel.py:296-353): the batches of
        `generator` (a Sequence - generator[i] -> inputs or (inputs, targets) - or any iterable of them) run with `in_flight`
        batches on the device at a time (Engine.in_flight: the next batch's backbone fills the CUs the current batch's small layers
        leave idle, + 15 % pages per second at batch 8); every batch's [boxes, scores, labels] are copied out as soon as THAT batch
        is done.  Returns the three arrays concatenated over the batches (Keras semantics).  RetinaNet.py's test() calls
        predict_on_batch page by page; a caller with many pages gets this instead."""
        if not self.bbox:
            raise ValueError("predict_generator: the inference model (retinanet_bbox) - the training model's outputs go through predict_on_batch")
        from collections import deque
        eng = self.engine()
        n = len(generator) if (steps is None and hasattr(generator, "__len__")) else steps
        it = (generator[i] for i in range(n)) if hasattr(generator, "__getitem__") and n is not None else iter(generator)
        outs, pending = ([], [], []), deque()

        def fetch():
            slot, views = pending.popleft()
            eng.wait_slot(slot)
            for o, v in zip(outs, views):
                o.append(_rt.host(v).copy())
        prev = eng.in_flight
        eng.join()
        eng.in_flight = max(1, int(in_flight))
        try:
            for k, item in enumerate(it):
                if n is not None and k >= n:
                    break
                x = item[0] if isinstance(item, (tuple, list)) else item
                x = _rt.dev(x, torch.float32) if not (isinstance(x, np.ndarray) and x.dtype == np.uint8) else _rt.dev(x, torch.uint8)
                if eng.in_flight == 1:
                    views = eng.detect(x)
                    torch.cuda.current_stream().synchronize()
                    for o, v in zip(outs, views):
                        o.append(_rt.host(v).copy())
                    continue
                if len(pending) == eng.in_flight:            # its buffer set is the one the next call rewrites
                    fetch()
                views = eng.detect(x)
                pending.append((eng.last_slot, views))
            while pending:
                fetch()
        finally:
            eng.join()
            eng.in_flight = prev
        return [np.concatenate(o, axis=0) if o else np.zeros((0,)) for o in outs]

    def train_on_batch(self, x, y):
        tr = self._get_trainer()
        total, reg, cls = tr.train_on_batch(_rt.dev(x, torch.float32), _rt.dev(y[0], torch.float32), _rt.dev(y[1], torch.float32),
                                            lr=getattr(self._compiled, "lr", None))
        return [total, reg, cls]

    def fit_generator(self, generator, steps_per_epoch=None, epochs=1, verbose=1, callbacks=None, validation_data=None,
                      workers=1, use_multiprocessing=False, max_queue_size=10, shuffle=True, initial_epoch=0, **kwargs):
        """ RetinaNet.py:280-291: Sequence-style generator (generator[i] -> (inputs, [regression_batch, labels_batch]))."""
        hist = History()
        callbacks = callbacks or []
        steps = steps_per_epoch or len(generator)
        if self._compiled is None:                                # fit without compile(): the defaults _get_trainer would take
            self._compiled = Adam(lr=1e-4, clipnorm=0.001)
        for cb in callbacks:
            if hasattr(cb, "set_model"):
                cb.set_model(self)
        for epoch in range(initial_epoch, epochs):
            for cb in callbacks:
                if hasattr(cb, "on_epoch_begin"):
                    cb.on_epoch_begin(epoch, {})
                if hasattr(cb, "schedule"):                       # LearningRateScheduler(lr_schedule), RetinaNet.py:47-66
                    self._compiled.lr = cb.schedule(epoch)
            sums = np.zeros(3)
            for i in range(steps):
                x, y = generator[i % len(generator)]
                sums += np.array(self.train_on_batch(x, y))
            logs = dict(zip(("loss", "regression_loss", "classification_loss"), (sums / steps).tolist()))
            if validation_data is not None:
                v = np.zeros(3)
                for j in range(len(validation_data)):
                    x, y = validation_data[j]
                    r, c = self.predict_on_batch(x) if not self.bbox else (None, None)
                    v += np.array([0.0, float(losses.smooth_l1()(y[0], r)), float(losses.focal()(y[1], c))])
                v /= max(1, len(validation_data))
                v[0] = v[1] + v[2]
                logs.update(dict(zip(("val_loss", "val_regression_loss", "val_classification_loss"), v.tolist())))
            for k, val in logs.items():
                hist.history.setdefault(k, []).append(val)
            if verbose:
                print("Epoch %d/%d - " % (epoch + 1, epochs) + " - ".join("%s: %.4f" % kv for kv in logs.items()))
            for cb in callbacks:
                if hasattr(cb, "on_epoch_end"):
                    cb.on_epoch_end(epoch, logs)
            if self.stop_training:
                break
        return hist

    # ---- weights
    def get_state(self):
        root = self._root()
        if root._trainer is not None:                               # pull the trained master weights back into Keras layout
            tr, eng = root._trainer, root._engine
            for name, lo in eng.layout.items():
                wm = tr.master[lo["woff"]:lo["woff"] + lo["rows"] * lo["K"]].view(lo["rows"], lo["K"]).cpu()
                cout = lo["cout"]
                if name == "conv1":
                    k = wm[:cout].reshape(cout, 8, 8, 4)[:, :7, :7, :3].permute(1, 2, 3, 0)
                else:
                    k = wm[:cout].reshape(cout, lo["kh"], lo["kw"], lo["cin"]).permute(1, 2, 3, 0)
                root._state[name + "/kernel"] = k.contiguous().numpy()
                if lo["has_bias"]:
                    root._state[name + "/bias"] = tr.master[tr.NW + lo["boff"]:tr.NW + lo["boff"] + cout].cpu().numpy()
        return root._state

    def save_weights(self, path):
        """Keras layer names and layouts either way: '*.h5' / '*.hdf5' -> a Keras save_weights() HDF5 file written by keras_h5.py
        (RetinaNet.py:70-79,153-163 save .h5), anything else -> .npz."""
        if str(path).endswith((".h5", ".hdf5")):
            _rt.keras_h5.save_keras_weights(path, self.get_state())
        else:
            np.savez(path, **self.get_state())

    save = save_weights

    def load_weights(self, path, by_name=True, skip_mismatch=False):
        root = self._root()
        if str(path).endswith((".h5", ".hdf5")):             # RetinaNet.py:320-340 loads training-*.h5 / inferModel.h5
            data = _rt.keras_h5.load_keras_state(path)
            names = list(data)
        else:
            data = np.load(path, allow_pickle=False)
            names = data.files
        for k in names:
            if k in root._state:
                if root._state[k].shape != data[k].shape:
                    if skip_mismatch:
                        continue
                    raise ValueError("shape mismatch for %s: %s vs %s" % (k, root._state[k].shape, data[k].shape))
                root._state[k] = data[k]
        if root._engine is not None:
            root._engine.load_state(root._state)
        root._trainer = None

    def summary(self):
        print("Model %s: %s, %d classes, %d anchors/cell" % (self.name, self.backbone, self.num_classes, self.num_anchors))


def load_model(path, custom_objects=None, backbone="resnet50", num_classes=1, num_anchors=9):
    m = Model(backbone, num_classes, num_anchors)
    m.load_weights(path)
    return m


class Backbone:
    """ Stores additional information on backbones (model/defineModel.py:9-48)."""

    def __init__(self, backbone):
        self.custom_objects = {
            'UpsampleLike': layers.UpsampleLike,
            'PriorProbability': initializers.PriorProbability,
            'RegressBoxes': layers.RegressBoxes,
            'FilterDetections': layers.FilterDetections,
            'Anchors': layers.Anchors,
            'ClipBoxes': layers.ClipBoxes,
            '_smooth_l1': losses.smooth_l1(),
            '_focal': losses.focal(),
        }
        self.backbone = backbone
        self.validate()

    def retinanet(self, *args, **kwargs):
        raise NotImplementedError('retinanet method not implemented.')

    def download_imagenet(self):
        raise NotImplementedError('download_imagenet method not implemented.')

    def validate(self):
        raise NotImplementedError('validate method not implemented.')

    def preprocess_image(self, inputs):
        raise NotImplementedError('preprocess_image method not implemented.')


class ResNetBackbone(Backbone):
    """ model/defineModel.py:50-75."""

    def retinanet(self, *args, **kwargs):
        return resnet_retinanet(*args, backbone=self.backbone, **kwargs)

    def validate(self):
        allowed_backbones = ['resnet50', 'resnet101', 'resnet152']
        backbone = self.backbone.split('_')[0]
        if backbone not in allowed_backbones:
            raise ValueError('Backbone (\'{}\') not in allowed backbones ({}).'.format(backbone, allowed_backbones))

    def preprocess_image(self, inputs):
        return utils.preprocess_image(inputs, mode='custom_tf')


def default_classification_model(num_classes, num_anchors, pyramid_feature_size=256, prior_probability=0.01,
                                 classification_feature_size=256, name='classification_submodel'):
    """ model/defineModel.py:78-125: descriptor of the 4 x (3x3, ReLU) + 3x3 -> sigmoid stack the engine executes."""
    return {"name": name, "prefix": "pyramid_classification", "outputs": num_classes * num_anchors, "activation": "sigmoid",
            "prior_probability": prior_probability, "feature_size": classification_feature_size}


def default_regression_model(num_values, num_anchors, pyramid_feature_size=256, regression_feature_size=256, name='regression_submodel'):
    """ model/defineModel.py:128-167."""
    return {"name": name, "prefix": "pyramid_regression", "outputs": num_values * num_anchors, "activation": None,
            "feature_size": regression_feature_size}


def retinanet(inputs, backbone_layers, num_classes, num_anchors=None, create_pyramid_features=None, submodels=None, name='retinanet',
              backbone='resnet50'):
    """ model/defineModel.py:232-267: outputs [regression, classification]."""
    if num_anchors is None:
        num_anchors = AnchorParameters_default.num_anchors()
    if create_pyramid_features is not None or submodels is not None:
        raise NotImplementedError("custom pyramid builders / submodels are not executed by the HIP engine")
    return Model(backbone, num_classes, num_anchors, name=name)


def retinanet_bbox(model=None, applyNms=True, class_specific_filter=True, name='retinanet-bbox', anchor_params=None, **kwargs):
    """ model/defineModel.py:296-353: appends Anchors -> RegressBoxes -> ClipBoxes -> FilterDetections; outputs
    [boxes, scores, labels]."""
    if anchor_params is None:
        anchor_params = AnchorParameters_default
    if model is None:
        model = retinanet(num_anchors=anchor_params.num_anchors(), **kwargs)
    else:
        utils.assert_training_model(model)
    if not applyNms or not class_specific_filter:
        raise NotImplementedError("only applyNms=True, class_specific_filter=True (the reference's use) runs on the device")
    return Model(model.backbone, model.num_classes, model.num_anchors, name=name, bbox=True, base=model, anchor_params=anchor_params)


def resnet_retinanet(num_classes, backbone='resnet50', inputs=None, modifier=None, **kwargs):
    """ model/defineModel.py:357-389."""
    if backbone not in ('resnet50', 'resnet101', 'resnet152'):
        raise ValueError('Backbone (\'{}\') is invalid.'.format(backbone))
    m = retinanet(inputs=inputs, num_classes=num_classes, backbone_layers=None, backbone=backbone, **kwargs)
    if modifier:
        m = modifier(m) or m
    return m


def resnet50_retinanet(num_classes, inputs=None, **kwargs):
    return resnet_retinanet(num_classes=num_classes, backbone='resnet50', inputs=inputs, **kwargs)


def resnet101_retinanet(num_classes, inputs=None, **kwargs):
    return resnet_retinanet(num_classes=num_classes, backbone='resnet101', inputs=inputs, **kwargs)


def resnet152_retinanet(num_classes, inputs=None, **kwargs):
    return resnet_retinanet(num_classes=num_classes, backbone='resnet152', inputs=inputs, **kwargs)
