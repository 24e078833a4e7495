import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'scratch'))
from ablate import run
for name, shape in [('layer1 32x32 64->64', (128, 32, 64, 64)), ('layer2 16x16 128->128', (128, 16, 128, 128)), ('layer3 8x8 256->256', (128, 8, 256, 256)), ('layer4 4x4 512->512', (128, 4, 512, 512)),
                    ('unet 16x16 128->64', (128, 16, 128, 64)), ('unet 8x8 256->128', (128, 8, 256, 128)), ('unet 4x4 512->256', (128, 4, 512, 256)), ('unet 2x2 512->512', (128, 2, 512, 512))]:
    out = []
    for tile in (6, 7, 8, 9, 3, 2, 1):
        try:
            us, tf = run(*shape, tile)
            out.append('t%d %.1fus' % (tile, us))
        except Exception as e:
            out.append('t%d -' % tile)
    print('%-24s %s' % (name, '  '.join(out)), flush=True)
