"""GPU: train.py -> checkpoint2model.py -> traverse_latent_space.py end to end on a tiny synthetic setup,
checking the reference's directory / file contract."""
import json
import os
import os.path as osp
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = osp.dirname(osp.dirname(osp.abspath(__file__)))


def run(cmd, cwd):
    env = dict(os.environ, PYTHONPATH=REPO)
    r = subprocess.run([sys.executable] + cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return r.stdout


def test_train_checkpoint_traverse_roundtrip(tmp_path):
    cwd = str(tmp_path)
    run([osp.join(REPO, 'train.py'), '--gan-type', 'StyleGAN2', '--stylegan2-resolution', '256', '-K', '4', '-D', '2',
         '--learn-gammas', '--max-iter', '4', '--batch-size', '2', '--log-freq', '2', '--ckp-freq', '2',
         '--random-init-generator', '--seed', '0'], cwd)
    exp = 'StyleGAN2-256-Z-ResNet-K4-D2-LearnGammas-eps0.25_0.45'
    wip = osp.join(cwd, 'experiments', 'wip', exp)
    done = osp.join(cwd, 'experiments', 'complete', exp)
    a = json.load(open(osp.join(wip, 'args.json')))
    assert a['gan_type'] == 'StyleGAN2' and a['num_support_sets'] == 4 and a['batch_size'] == 2 and 'seed' not in a
    assert set(a) == {'gan_type', 'z_truncation', 'biggan_target_classes', 'stylegan2_resolution', 'shift_in_w_space',
                      'num_support_sets', 'num_support_dipoles', 'learn_alphas', 'learn_gammas', 'gamma', 'support_set_lr',
                      'reconstructor_type', 'min_shift_magnitude', 'max_shift_magnitude', 'reconstructor_lr', 'max_iter',
                      'batch_size', 'lambda_cls', 'lambda_reg', 'log_freq', 'ckp_freq', 'tensorboard', 'cuda'}
    stats = json.load(open(osp.join(wip, 'stats.json')))
    assert set(stats) == {'2', '4'} and set(stats['2']) == {'accuracy', 'classification_loss', 'regression_loss', 'total_loss'}
    for f in ('support_sets_init.pt', 'support_sets.pt', 'reconstructor.pt', 'checkpoint.pt'):
        assert osp.isfile(osp.join(wip, 'models', f)), f
    assert osp.isfile(osp.join(done, 'models', 'support_sets.pt')) and not osp.exists(osp.join(done, 'models', 'checkpoint.pt'))
    ck = torch.load(osp.join(wip, 'models', 'checkpoint.pt'), map_location='cpu')
    # the reference's three keys (lib/trainer.py:290-295) + the optional Adam-moment key its readers never look at
    assert set(ck) == {'iter', 'support_sets', 'reconstructor', 'optim'} and ck['iter'] == 4
    assert ck['optim']['step'] == 4 and ck['optim']['exp_avg'].numel() == ck['optim']['exp_avg_sq'].numel() > 0
    assert float(ck['optim']['exp_avg_sq'].max()) > 0
    # a checkpoint in the REFERENCE's layout (no 'optim') resumes as well: optimizers restart from zero moments, like the reference
    ref_like = {k: ck[k] for k in ('iter', 'support_sets', 'reconstructor')}
    ss = ck['support_sets']
    assert ss['SUPPORT_SETS'].shape == (4, 2 * 2 * 512) and ss['ALPHAS'].shape == (4, 4) and ss['LOGGAMMA'].shape == (4, 1)
    r = ck['reconstructor']
    assert r['features_extractor.conv1.weight'].shape == (64, 6, 7, 7) and r['features_extractor.conv1.weight'].is_contiguous()
    assert r['features_extractor.fc.weight'].shape == (1000, 512) and r['path_indices.weight'].shape == (4, 512)
    init = torch.load(osp.join(wip, 'models', 'support_sets_init.pt'), map_location='cpu')
    assert not torch.equal(init['SUPPORT_SETS'], ss['SUPPORT_SETS'])            # training moved the support sets
    run([osp.join(REPO, 'checkpoint2model.py'), '--exp', wip], cwd)
    assert osp.isfile(osp.join(wip, 'models', 'support_sets-4.pt')) and osp.isfile(osp.join(wip, 'models', 'reconstructor-4.pt'))
    # latent-code pool in the reference's layout: experiments/latent_codes/<gan>/<pool>/<hash>/latent_code.pt
    for j, h in enumerate(('aaaa', 'bbbb')):
        d = osp.join(cwd, 'experiments', 'latent_codes', 'StyleGAN2', 'pool2', h)
        os.makedirs(d)
        torch.save(torch.randn(1, 512, generator=torch.Generator().manual_seed(j)), osp.join(d, 'latent_code.pt'))
    run([osp.join(REPO, 'traverse_latent_space.py'), '--exp', done, '--pool', 'pool2', '--shift-steps', '4', '--eps', '0.2',
         '--shift-leap', '2', '--img-size', '64', '--random-init-generator'], cwd)
    out = osp.join(done, 'results', 'pool2', '8_0.2_1.6')
    for h in ('aaaa', 'bbbb'):
        plc = torch.load(osp.join(out, h, 'paths_latent_codes.pt'))
        assert plc.shape == (4, 5, 512)                                        # [K, 2*steps/leap + 1, d]
        assert osp.isfile(osp.join(out, h, 'original_image.jpg'))
        assert sorted(os.listdir(osp.join(out, h, 'paths_images', 'path_003'))) == ['%06d.jpg' % t for t in range(5)]
    # resume: continue the finished experiment to iteration 6, once from our checkpoint (moments restored) and once from a
    # reference-layout checkpoint
    common = ['--gan-type', 'StyleGAN2', '--stylegan2-resolution', '256', '-K', '4', '-D', '2', '--learn-gammas', '--batch-size', '2',
              '--log-freq', '2', '--ckp-freq', '2', '--random-init-generator', '--seed', '0']
    out1 = run([osp.join(REPO, 'train.py')] + common + ['--max-iter', '6'], cwd)
    # with the moments in the checkpoint the state is the one AFTER iteration 4: the run continues at 5 and the Adam step count
    # stays equal to the iteration number (a reference-layout checkpoint re-runs `iter` with fresh optimizers, as the reference does)
    assert 'Restored Adam moments (step 4)' in out1 and 'Start training from iteration 5' in out1
    ck6 = torch.load(osp.join(wip, 'models', 'checkpoint.pt'), map_location='cpu')
    assert ck6['iter'] == 6 and ck6['optim']['step'] == 6
    torch.save(dict(ref_like, iter=6), osp.join(wip, 'models', 'checkpoint.pt'))
    out2 = run([osp.join(REPO, 'train.py')] + common + ['--max-iter', '8'], cwd)
    assert 'Restored Adam moments' not in out2 and 'Start training from iteration 6' in out2
    z0 = torch.load(osp.join(cwd, 'experiments', 'latent_codes', 'StyleGAN2', 'pool2', 'aaaa', 'latent_code.pt'))
    assert torch.allclose(plc.new_tensor(torch.load(osp.join(out, 'aaaa', 'paths_latent_codes.pt'))[0, 2]), z0[0], atol=1e-6)


def test_sample_gan_pool_layout_feeds_traversal(tmp_path):
    """sample_gan.py writes the reference's pool layout (sample_gan.py:70-91,156-179) and traverse_latent_space.py reads it."""
    from hashlib import sha1
    cwd = str(tmp_path)
    run([osp.join(REPO, 'sample_gan.py'), '-v', '--gan-type', 'StyleGAN2', '--stylegan2-resolution', '256', '--num-samples', '3',
         '--pool', 'p3', '--z-truncation', '0.7', '--random-init-generator', '--seed', '1', '--batch-size', '2'], cwd)
    pool = osp.join(cwd, 'experiments', 'latent_codes', 'StyleGAN2', 'p3')
    a = json.load(open(osp.join(pool, 'args.json')))
    assert set(a) == {'verbose', 'gan_type', 'shift_in_w_space', 'z_truncation', 'biggan_target_classes', 'stylegan2_resolution',
                      'num_samples', 'pool', 'cuda'} and a['z_truncation'] == 0.7
    dirs = sorted(d for d in os.listdir(pool) if osp.isdir(osp.join(pool, d)))
    assert len(dirs) == 3
    from PIL import Image
    for d in dirs:
        z = torch.load(osp.join(pool, d, 'latent_code.pt'))
        assert z.shape == (1, 512) and z.dtype == torch.float32 and float(z.abs().max()) <= 0.7
        assert sha1(z.numpy()).hexdigest() == d
        assert Image.open(osp.join(pool, d, 'image.jpg')).size == (256, 256)
    # default pool name <gan_type>_<num_samples>; BigGAN needs its classes and names the directory after them
    run([osp.join(REPO, 'sample_gan.py'), '--gan-type', 'SNGAN_MNIST', '--num-samples', '2', '--random-init-generator'], cwd)
    assert len([d for d in os.listdir(osp.join(cwd, 'experiments', 'latent_codes', 'SNGAN_MNIST', 'SNGAN_MNIST_2')) if d != 'args.json']) == 2
    run([osp.join(REPO, 'sample_gan.py'), '--gan-type', 'BigGAN', '--biggan-target-classes', '14', '239', '--num-samples', '1',
         '--random-init-generator'], cwd)
    assert osp.isdir(osp.join(cwd, 'experiments', 'latent_codes', 'BigGAN-14-239', 'BigGAN-14-239_1'))
