"""On-disk dataset directories of 3d-beats, read side only.

Layout written by the reference's data generators (src/live_data_convert.py:287-298, 456-458) and read by
`DecisionTreeDatasetConfig` (src/decision_tree.py:21-122):

    <dir>/config.json            {"img_dims": [W, H], "num_images": N, "id_to_color": {"0": [0,0,0,0], "1": [r,g,b,255], ...}}
    <dir>/00000000_depth.png     16-bit depth, background already 65535
    <dir>/00000000_labels.png    16-bit class ids (0 = unlabelled)

This class keeps the reference's constructor, attributes and block getters so that harnesses such as
src/test_on_saved_model.py:44-56 run unchanged, but blocks are plain device arrays filled from the PNGs
(the reference stages them through nvcomp-compressed blocks, which is training-side machinery).
"""
import json
import os

import numpy as np


class DecisionTreeDatasetConfig:
    @staticmethod
    def multiple(dataset_dir, images):
        """Several datasets over one directory, e.g. [(n_train, per_block, 'train'), (n_test, None, 'test')]
        (decision_tree.py:24-44).  Like the reference, every part draws its own random subset."""
        with open(os.path.join(dataset_dir, 'config.json')) as fh:
            total_images = json.loads(fh.read())['num_images']
        assert sum(n for n, _, _ in images) <= total_images
        return tuple(DecisionTreeDatasetConfig(dataset_dir, num_images=n, images_per_block=(per_block or n),
                                               imgs_name=name) for n, per_block, name in images)

    def __init__(self, dataset_dir, num_images=0, images_per_block=0, imgs_name='data0', shuffle=True):
        self.dataset_dir = dataset_dir
        with open(os.path.join(dataset_dir, 'config.json')) as fh:
            cfg = json.loads(fh.read())
        self.cfg = cfg
        self.imgs_name = imgs_name

        self.img_dims = tuple(cfg['img_dims'])  # (x, y)
        self.id_to_color = {0: np.array([0, 0, 0, 0], dtype=np.uint8)}
        for i, c in cfg['id_to_color'].items():
            self.id_to_color[int(i)] = np.array(c, dtype=np.uint8)

        self.total_available_images = cfg['num_images']
        self.num_images = num_images
        if self.num_images == 0:
            return
        assert self.num_images <= self.total_available_images

        self.images_per_block = images_per_block or self.num_images
        assert self.num_images % self.images_per_block == 0
        self.num_image_blocks = self.num_images // self.images_per_block

        idxes = list(range(self.total_available_images))
        if shuffle:
            np.random.shuffle(idxes)   # the reference draws from the global numpy RNG (decision_tree.py:68)
        self.img_idxes = idxes[0:self.num_images]

    # -- reading -----------------------------------------------------------------------------
    def _load(self, img_idx, name):
        from PIL import Image
        path = os.path.join(self.dataset_dir, f'{str(img_idx).zfill(8)}_{name}.png')
        a = np.array(Image.open(path)).astype(np.uint16)
        assert a.shape == (self.img_dims[1], self.img_dims[0]), f'{path}: {a.shape}'
        return a

    def get_block_cpu(self, block_num, name):
        out = np.empty((self.images_per_block, self.img_dims[1], self.img_dims[0]), dtype=np.uint16)
        for j in range(self.images_per_block):
            out[j] = self._load(self.img_idxes[block_num * self.images_per_block + j], name)
        return out

    def get_depth_block_cu(self, block_num, arr_out):
        arr_out.set(self.get_block_cpu(block_num, 'depth'))

    def get_labels_block_cu(self, block_num, arr_out):
        arr_out.set(self.get_block_cpu(block_num, 'labels'))

    # -- bookkeeping (decision_tree.py:85-122) -------------------------------------------------
    def num_classes(self):
        return len(self.id_to_color)

    def num_pixels(self):
        return self.num_images * self.img_dims[0] * self.img_dims[1]

    def images_shape(self):
        return (self.num_images, self.img_dims[1], self.img_dims[0])

    def _palette(self):
        """(sorted RGBA keys as uint32, their class ids, id -> RGBA table) built once from id_to_color."""
        if getattr(self, "_pal", None) is None:
            ids = np.array(sorted(self.id_to_color), dtype=np.int64)
            rgba = np.stack([np.asarray(self.id_to_color[int(i)], dtype=np.uint8) for i in ids])
            keys = np.ascontiguousarray(rgba).view(np.uint32).reshape(-1)
            order = np.argsort(keys, kind="stable")
            table = np.zeros((int(ids.max()) + 1, 4), dtype=np.uint8)
            table[ids] = rgba
            self._pal = (keys[order], ids[order], table)
        return self._pal

    def convert_colors_to_ids(self, labels_color):
        """RGBA label image [H, W, 4] -> class ids [H, W] (decision_tree.py:97-110): every pixel must carry one of the
        dataset's colours.  One 32-bit key per pixel, looked up in the sorted palette."""
        keys, ids, _ = self._palette()
        px = np.ascontiguousarray(labels_color, dtype=np.uint8)
        assert px.shape == (self.img_dims[1], self.img_dims[0], 4), px.shape
        k = px.view(np.uint32).reshape(px.shape[0], px.shape[1])
        at = np.minimum(np.searchsorted(keys, k), keys.size - 1)
        assert np.array_equal(keys[at], k), 'every pixel must carry a known label colour'
        return ids[at].astype(np.uint16)

    def convert_ids_to_colors(self, labels_ids):
        """Class ids [N, H, W] -> RGBA [N, H, W, 4] (decision_tree.py:112-122); an id the dataset does not know stays
        transparent black."""
        n, y, x = labels_ids.shape
        assert (y, x) == (self.img_dims[1], self.img_dims[0])
        _, _, table = self._palette()
        ids = np.asarray(labels_ids).astype(np.int64)
        known = (ids >= 0) & (ids < table.shape[0])
        return np.where(known[..., None], table[np.where(known, ids, 0)], np.uint8(0)).astype(np.uint8)


def write_dataset(dataset_dir, depth, labels, id_to_color):
    """Writes a dataset directory in the layout above (used by tests and tools; the reference's writer is
    its GL data generator).  depth, labels: uint16 [N, H, W]; id_to_color: {id: [r, g, b, a]} without id 0."""
    from PIL import Image
    os.makedirs(dataset_dir, exist_ok=True)
    n, h, w = depth.shape
    cfg = {'img_dims': [int(w), int(h)], 'num_images': int(n), 'id_to_color': {'0': [0, 0, 0, 0]}}
    for k, c in id_to_color.items():
        cfg['id_to_color'][str(int(k))] = [int(v) for v in c]
    with open(os.path.join(dataset_dir, 'config.json'), 'w') as fh:
        fh.write(json.dumps(cfg))
    for i in range(n):
        Image.fromarray(np.ascontiguousarray(depth[i], dtype=np.uint16)).save(
            os.path.join(dataset_dir, f'{str(i).zfill(8)}_depth.png'))
        Image.fromarray(np.ascontiguousarray(labels[i], dtype=np.uint16)).save(
            os.path.join(dataset_dir, f'{str(i).zfill(8)}_labels.png'))


def evaluate_saved_model(model_path, dataset_dir, num_images, out_dir=None):
    """The accuracy harness of src/test_on_saved_model.py:23-67 without the GLFW window: returns
    `pct. matching pixels` = matches / labelled pixels, optionally saving colour renders."""
    from .decision_tree import DecisionForest, DecisionTreeEvaluator
    from .device import DeviceArray
    from .util import MAX_UINT16
    forest = DecisionForest.load(model_path)
    ev = DecisionTreeEvaluator()
    ds = DecisionTreeDatasetConfig(dataset_dir, num_images=num_images, imgs_name='test')
    depth = DeviceArray(ds.images_shape(), np.uint16)
    ds.get_depth_block_cu(0, depth)
    truth = DeviceArray(ds.images_shape(), np.uint16)
    ds.get_labels_block_cu(0, truth)
    truth_cpu = truth.get()
    out = DeviceArray(ds.images_shape(), np.uint16).fill(MAX_UINT16)
    ev.get_labels_forest(forest, depth, out)
    got = out.get()
    pct = float(np.sum(got == truth_cpu) / np.sum(truth_cpu > 0))
    if out_dir:
        from PIL import Image
        os.makedirs(out_dir, exist_ok=True)
        render = ds.convert_ids_to_colors(got)
        for i in range(ds.num_images):
            Image.fromarray(render[i]).save(os.path.join(out_dir, f'eval_labels_{str(i).zfill(8)}.png'))
    return pct


def train_forest(dataset_dir, num_train, num_test, proposals, proposals_block, out_trees, max_depth, out_path=None,
                 trees_to_try=None, train_block=None, log=print, tree_seed=None, group=None):
    """The flow of src/train_model.py:52-139 without its GLFW window: train `trees_to_try` candidate trees, keep
    the `out_trees` best by test accuracy, evaluate the forest, save it as the reference's .npy.

    tree_seed (not in the reference): candidate tree i draws its proposals from numpy.random.seed(tree_seed + i)
    instead of continuing the global stream.  That makes the candidates independent of each other, which is what
    lets them be trained on different GPUs: with torch.distributed initialised (one process per GPU), rank r trains
    candidates r, r + world, ...; the trees and their scores are exchanged with all_gather_object and every rank
    keeps the same best ones -- the same forest, bit for bit, as one process with the same tree_seed."""
    from .decision_tree import DecisionForest, DecisionTree, DecisionTreeEvaluator, DecisionTreeTrainer
    from .device import DeviceArray
    from .util import MAX_UINT16
    trees_to_try = trees_to_try or out_trees
    if tree_seed is not None:
        np.random.seed(int(tree_seed))     # the random train/test split: the same in every process
    train_data, test_data = DecisionTreeDatasetConfig.multiple(dataset_dir, [(num_train, train_block, 'train'),
                                                                             (num_test, None, 'test')])
    trainer = DecisionTreeTrainer(train_block or num_train, proposals_block)
    evaluator = DecisionTreeEvaluator()
    tree1 = DecisionTree(max_depth, train_data.num_classes())
    trainer.allocate(train_data, proposals, tree1.max_depth)
    out_cu = DeviceArray(test_data.images_shape(), np.uint16)
    test_depth, test_labels = DeviceArray(test_data.images_shape(), np.uint16), DeviceArray(test_data.images_shape(), np.uint16)
    test_data.get_depth_block_cu(0, test_depth)
    test_data.get_labels_block_cu(0, test_labels)
    truth = test_labels.get()
    best = [None] * out_trees
    forest_cpu = np.zeros((out_trees, tree1.TOTAL_TREE_NODES, tree1.TREE_NODE_ELS), dtype=np.float32)
    rank, world = 0, 1
    if tree_seed is not None:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            rank, world = dist.get_rank(group), dist.get_world_size(group)
    candidates = {}
    for i in range(rank, trees_to_try, world):
        if tree_seed is not None:
            np.random.seed(int(tree_seed) + i)
        trainer.train(train_data, tree1)
        out_cu.fill(MAX_UINT16)
        evaluator.get_labels(tree1, test_depth, out_cu)
        pct = float(np.sum(out_cu.get() == truth) / np.sum(truth > 0))
        log(f'tree {i}: pct. matching pixels: {pct:.4f}')
        candidates[i] = (pct, tree1.tree_out_cu.get())
    if world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, candidates, group=group)
        candidates = {i: v for part in parts for i, v in part.items()}
    for i in range(trees_to_try):        # the reference's selection, in candidate order
        pct, tree_np = candidates[i]
        slot = -1
        if None in best:
            slot = best.index(None)
        elif pct > min(best):
            slot = best.index(min(best))
        if slot > -1:
            best[slot] = pct
            forest_cpu[slot] = tree_np
    forest = DecisionForest(out_trees, max_depth, test_data.num_classes())
    forest.forest_cu.set(forest_cpu)
    out_cu.fill(MAX_UINT16)
    evaluator.get_labels_forest(forest, test_depth, out_cu)
    pct = float(np.sum(out_cu.get() == truth) / np.sum(truth > 0))
    log(f'FOREST pct. matching pixels: {pct:.4f}')
    if out_path:
        np.save(out_path, forest_cpu)
    return forest_cpu, pct
