"""Class-name tables of the ImageNet VID (30 classes + background) and DET (200 classes +
background) tasks and the VID -> DET index map.

Mirror of the reference's vdet/dataset.py:7-19, which reads misc/imagenet_vdet_classes.txt and
misc/imagenet_det_200_classes.txt at import time; here the (public ILSVRC) names are embedded so
the package has no data-file dependency.  Index 0 is '__background__' in both lists.
"""

imagenet_vdet_classes = [
    '__background__', 'airplane', 'antelope', 'bear', 'bicycle', 'bird', 'bus', 'car', 'cattle',
    'dog', 'domestic_cat', 'elephant', 'fox', 'giant_panda', 'hamster', 'horse', 'lion',
    'lizard', 'monkey', 'motorcycle', 'rabbit', 'red_panda', 'sheep', 'snake', 'squirrel',
    'tiger', 'train', 'turtle', 'watercraft', 'whale', 'zebra',
]

imagenet_vdet_class_idx = dict(zip(imagenet_vdet_classes, range(len(imagenet_vdet_classes))))

imagenet_det_200_classes = [
    '__background__', 'accordion', 'airplane', 'ant', 'antelope', 'apple', 'armadillo',
    'artichoke', 'axe', 'baby_bed', 'backpack', 'bagel', 'balance_beam', 'banana', 'band_aid',
    'banjo', 'baseball', 'basketball', 'bathing_cap', 'beaker', 'bear', 'bee', 'bell_pepper',
    'bench', 'bicycle', 'binder', 'bird', 'bookshelf', 'bow', 'bow_tie', 'bowl', 'brassiere',
    'burrito', 'bus', 'butterfly', 'camel', 'can_opener', 'car', 'cart', 'cattle', 'cello',
    'centipede', 'chain_saw', 'chair', 'chime', 'cocktail_shaker', 'coffee_maker',
    'computer_keyboard', 'computer_mouse', 'corkscrew', 'cream', 'croquet_ball', 'crutch',
    'cucumber', 'cup_or_mug', 'diaper', 'digital_clock', 'dishwasher', 'dog', 'domestic_cat',
    'dragonfly', 'drum', 'dumbbell', 'electric_fan', 'elephant', 'face_powder', 'fig',
    'filing_cabinet', 'flower_pot', 'flute', 'fox', 'french_horn', 'frog', 'frying_pan',
    'giant_panda', 'goldfish', 'golf_ball', 'golfcart', 'guacamole', 'guitar', 'hair_dryer',
    'hair_spray', 'hamburger', 'hammer', 'hamster', 'harmonica', 'harp', 'hat_with_a_wide_brim',
    'head_cabbage', 'helmet', 'hippopotamus', 'horizontal_bar', 'horse', 'hotdog', 'iPod',
    'isopod', 'jellyfish', 'koala_bear', 'ladle', 'ladybug', 'lamp', 'laptop', 'lemon', 'lion',
    'lipstick', 'lizard', 'lobster', 'maillot', 'maraca', 'microphone', 'microwave', 'milk_can',
    'miniskirt', 'monkey', 'motorcycle', 'mushroom', 'nail', 'neck_brace', 'oboe', 'orange',
    'otter', 'pencil_box', 'pencil_sharpener', 'perfume', 'person', 'piano', 'pineapple',
    'ping-pong_ball', 'pitcher', 'pizza', 'plastic_bag', 'plate_rack', 'pomegranate',
    'popsicle', 'porcupine', 'power_drill', 'pretzel', 'printer', 'puck', 'punching_bag',
    'purse', 'rabbit', 'racket', 'ray', 'red_panda', 'refrigerator', 'remote_control',
    'rubber_eraser', 'rugby_ball', 'ruler', 'salt_or_pepper_shaker', 'saxophone', 'scorpion',
    'screwdriver', 'seal', 'sheep', 'ski', 'skunk', 'snail', 'snake', 'snowmobile', 'snowplow',
    'soap_dispenser', 'soccer_ball', 'sofa', 'spatula', 'squirrel', 'starfish', 'stethoscope',
    'stove', 'strainer', 'strawberry', 'stretcher', 'sunglasses', 'swimming_trunks', 'swine',
    'syringe', 'table', 'tape_player', 'tennis_ball', 'tick', 'tie', 'tiger', 'toaster',
    'traffic_light', 'train', 'trombone', 'trumpet', 'turtle', 'tv_or_monitor', 'unicycle',
    'vacuum', 'violin', 'volleyball', 'waffle_iron', 'washer', 'water_bottle', 'watercraft',
    'whale', 'wine_bottle', 'zebra',
]

imagenet_det_200_class_idx = dict(zip(imagenet_det_200_classes, range(len(imagenet_det_200_classes))))

# VID class index -> DET class index (same class name in both lists)
index_vdet_to_det = dict((imagenet_vdet_classes.index(name), imagenet_det_200_classes.index(name))
                         for name in imagenet_vdet_classes)
