"""10-fold cross-validation driver on the HIP path.

Restates the experiment loop of /root/reference/train.py:69-148 -- same flags (``--data_type --batch_size
--num_epochs --seed``, train.py:17-25), a fresh ``Model`` + Adam per fold (train.py:97-99), the shipped
fold index files (train.py:103-106), per-epoch train/test loss and accuracy (train.py:115-120), a
``.pth`` state_dict per fold (train.py:129), per-fold and overall CSVs (train.py:130-131,144-145) and
the final mean/std line (train.py:146-148) -- without visdom/pandas/tqdm/PyG, which cannot travel to the
GPU box.  ``--synthetic N`` trains on N synthetic graphs of the named shape when the TU files are absent.

    python train.py --data_type MUTAG --data_root data          # TU files under data/MUTAG/raw
    python train.py --data_type COLLAB --synthetic 500 --num_epochs 3 --folds 2
"""
from __future__ import annotations

import argparse
import csv
import os
import random
from typing import List

import numpy as np
import torch

from . import synth
from .model import Model
from .train import Trainer
from .device_data import DeviceDataset, DeviceLoader
from .tudataset import GraphLoader, TUData, make_fold_indices, read_fold_indices, read_tu_dataset

CHOICES = ['DD', 'PTC_MR', 'NCI1', 'PROTEINS', 'IMDB-BINARY', 'IMDB-MULTI', 'MUTAG', 'COLLAB']
SYNTH_SHAPE = {'DD': 'DD', 'PROTEINS': 'PROTEINS', 'MUTAG': 'MUTAG', 'COLLAB': 'COLLAB', 'IMDB-BINARY': 'IMDB',
               'IMDB-MULTI': 'IMDB', 'PTC_MR': 'MUTAG', 'NCI1': 'MUTAG'}


def get_args(argv=None):
    p = argparse.ArgumentParser(description='Train Model (MI355X-native DGCNN)')
    p.add_argument('--data_type', default='DD', type=str, choices=CHOICES, help='dataset type')
    p.add_argument('--batch_size', default=50, type=int, help='train batch size')
    p.add_argument('--num_epochs', default=100, type=int, help='train epochs number')
    p.add_argument('--seed', default=324, type=int, help='random seed')
    p.add_argument('--data_root', default='data', type=str, help='directory holding <data_type>/raw/*.txt and 10fold_idx')
    p.add_argument('--synthetic', default=0, type=int,
                   help='use N synthetic graphs of that shape (class = edge-density level) instead of TU files')
    p.add_argument('--folds', default=10, type=int, help='number of folds to run (<= 10)')
    p.add_argument('--out_dir', default='.', type=str, help='where epochs/ and statistics/ are written')
    p.add_argument('--device', default='cuda', type=str)
    p.add_argument('--host_loader', dest='device_loader', action='store_false',
                   help='collate every batch on the host like the reference DataLoader (default: dataset resident in HBM, '
                        'batches assembled on the device)')
    p.add_argument('--exclusive-device', dest='exclusive_device', action='store_true',
                   help='promise that nothing else runs on this GPU (admits the step form whose launch also carries the next '
                        "batch's whole graph preparation; Trainer(exclusive_device=True))")
    return p.parse_args(argv)


def set_determ(seed: int) -> None:
    """Seeding of /root/reference/set_determ.py:11-30 (python, torch, numpy)."""
    random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)


def load_dataset(opt) -> TUData:
    if opt.synthetic > 0:
        shape = synth.SHAPES[SYNTH_SHAPE[opt.data_type]]
        graphs = synth.make_graphs(shape.name, opt.synthetic, seed=opt.seed, labels="structure")   # learnable
        return TUData(graphs, shape.num_classes, opt.data_type)
    return read_tu_dataset(os.path.join(opt.data_root, opt.data_type), opt.data_type, use_node_attr=True)


def run(opt) -> dict:
    set_determ(opt.seed)
    data_set = load_dataset(opt)
    print(f'{data_set.num_features=}, {data_set.num_classes=}')
    os.makedirs(os.path.join(opt.out_dir, 'epochs'), exist_ok=True)
    os.makedirs(os.path.join(opt.out_dir, 'statistics'), exist_ok=True)
    over = {'train_accuracy': [], 'test_accuracy': []}
    gen = torch.Generator().manual_seed(opt.seed)
    dev_set = DeviceDataset(data_set, opt.device) if (opt.device_loader and str(opt.device).startswith('cuda')) else None
    for fold in range(1, opt.folds + 1):
        model = Model(data_set.num_features, data_set.num_classes).to(opt.device)
        trainer = Trainer(model, exclusive_device=opt.exclusive_device)      # Adam defaults, as Adam(model.parameters()) at train.py:99
        idx_dir = os.path.join(opt.data_root, opt.data_type)
        if opt.synthetic == 0 and os.path.isdir(os.path.join(idx_dir, '10fold_idx')):
            tr_idx, te_idx = read_fold_indices(idx_dir, fold)
        else:
            tr_idx, te_idx = make_fold_indices(len(data_set), fold, max(opt.folds, 2), opt.seed)
        if dev_set is not None:       # dataset resident in HBM, batches assembled by one kernel launch each
            train_loader = DeviceLoader(dev_set, opt.batch_size, tr_idx, shuffle=True, generator=gen)
            test_loader = DeviceLoader(dev_set, opt.batch_size, te_idx, shuffle=False)
        else:
            train_loader = GraphLoader(data_set[tr_idx], opt.batch_size, shuffle=True, generator=gen, device=opt.device)
            test_loader = GraphLoader(data_set[te_idx], opt.batch_size, shuffle=False, device=opt.device)
        res = {'train_loss': [], 'test_loss': [], 'train_accuracy': [], 'test_accuracy': []}
        for epoch in range(1, opt.num_epochs + 1):
            tl, ta = trainer.train_epoch(train_loader, train_loader.num_samples)
            vl, va = trainer.test_epoch(test_loader, test_loader.num_samples)
            res['train_loss'].append(tl); res['train_accuracy'].append(ta)
            res['test_loss'].append(vl); res['test_accuracy'].append(va)
        torch.save(model.state_dict(), os.path.join(opt.out_dir, 'epochs', f'{opt.data_type}_{fold}.pth'))
        with open(os.path.join(opt.out_dir, 'statistics', f'{opt.data_type}_results_{fold}.csv'), 'w', newline='') as f:
            wr = csv.writer(f)
            wr.writerow(['epoch', 'train_loss', 'test_loss', 'train_accuracy', 'test_accuracy'])
            for e in range(opt.num_epochs):
                wr.writerow([e + 1, res['train_loss'][e], res['test_loss'][e], res['train_accuracy'][e], res['test_accuracy'][e]])
        over['train_accuracy'].append(res['train_accuracy'][-1])
        over['test_accuracy'].append(res['test_accuracy'][-1])
        print(f'[{fold}] Train Acc: {res["train_accuracy"][-1]:.2f}% Test Acc: {res["test_accuracy"][-1]:.2f}%')
    with open(os.path.join(opt.out_dir, 'statistics', f'{opt.data_type}_results_overall.csv'), 'w', newline='') as f:
        wr = csv.writer(f)
        wr.writerow(['fold', 'train_accuracy', 'test_accuracy'])
        for k in range(len(over['train_accuracy'])):
            wr.writerow([k + 1, over['train_accuracy'][k], over['test_accuracy'][k]])
    print('Overall Training Accuracy: %.2f%% (std: %.2f) Testing Accuracy: %.2f%% (std: %.2f)' %
          (np.array(over['train_accuracy']).mean(), np.array(over['train_accuracy']).std(),
           np.array(over['test_accuracy']).mean(), np.array(over['test_accuracy']).std()))
    return over


def main(argv=None):
    return run(get_args(argv))


if __name__ == '__main__':
    main()
