"""Checkpoint I/O with the reference's file layout (src/utils.py:139-195): a dict with keys
{'epoch', 'valid_loss_min', 'state_dict', 'optimizer'} written to model_temporary / model_{epoch} / model_most_recent /
model_best .ckpt.  `state_dict` has the reference's keys and shapes and `optimizer` uses
torch.optim.AdamW's state_dict format, so files interchange with the reference in both directions."""
import os

import torch


def save_ckp(epoch, valid_loss, model, optimizer, is_best, checkpoint_path, temporary_save=False, epoch_save=False):
    """src/utils.py:139-175."""
    if not os.path.exists(checkpoint_path):
        os.makedirs(checkpoint_path)
    from .engine import join_streams
    join_streams()                                 # a deferred discriminator phase may still be updating its range
    state = {
        'epoch': epoch + 1,
        'valid_loss_min': valid_loss,
        'state_dict': {k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()},
        'optimizer': optimizer.state_dict(),
    }
    if temporary_save:
        torch.save(state, checkpoint_path + '/model_temporary.ckpt')
        return
    if epoch_save:
        torch.save(state, checkpoint_path + f'/model_{epoch}.ckpt')
        return
    torch.save(state, checkpoint_path + '/model_most_recent.ckpt')
    if is_best:
        torch.save(state, checkpoint_path + '/model_best.ckpt')


def load_ckp(checkpoint_fpath, model, optimizer):
    """src/utils.py:178-195."""
    if not os.path.exists(checkpoint_fpath):
        raise Exception("There is no model at the desired checkpoint")
    checkpoint = torch.load(checkpoint_fpath, map_location="cpu", weights_only=False)
    model.load_state_dict(checkpoint['state_dict'])
    optimizer.load_state_dict(checkpoint['optimizer'])
    return checkpoint['epoch'], checkpoint['valid_loss_min'], model, optimizer
