#!/usr/bin/env python
"""python train_s2.py --save_dir DIR --batch_size B --lr_h LR --num_epoch N --weight_loss_* ...   (source/train_s2.py __main__)"""
import _train_main

if __name__ == '__main__':
    _train_main.main('s2')
