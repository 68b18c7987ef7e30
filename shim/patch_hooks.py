#!/usr/bin/env python3
"""Generates, at build time, copies of four MagickCore sources with the accelerate call
sites the reference does not have (or has commented out) switched in — SURVEY 8b: "new hooks
for Morphology and Colorspace", the disabled UnsharpMask stanza, the caller-less
ContrastStretch, WaveletDenoise's hook without its softness argument.  Each hook is the reference's own three-line idiom
(effect.c:783-787).  The copies are written under shim/_build/ (never committed, never
shipped); the reference tree is only read.

    python shim/patch_hooks.py <reference MagickCore dir> <output dir>
"""
import os
import sys

# entry points of shim/accelerate_hip.c that accelerate-private.h does not declare
MORPHOLOGY_PROTOTYPE = '''
#if defined(MAGICKCORE_OPENCL_SUPPORT)
extern MagickPrivate Image *AccelerateMorphologyApply(const Image *,const MorphologyMethod,
  const ssize_t,const KernelInfo *,const CompositeOperator,const double,ExceptionInfo *);
#endif
'''
WAVELET_PROTOTYPE = '''
#if defined(MAGICKCORE_OPENCL_SUPPORT)
extern MagickPrivate Image *AccelerateWaveletDenoiseImageSoft(const Image *,const double,
  const double,ExceptionInfo *);
#endif
'''
COLORSPACE_PROTOTYPE = '''
#if defined(MAGICKCORE_OPENCL_SUPPORT)
extern MagickPrivate MagickBooleanType AccelerateTransformImageColorspace(Image *,
  const ColorspaceType,ExceptionInfo *);
#endif
'''


def once(text, anchor, replacement, name):
    if text.count(anchor) != 1:
        raise SystemExit("%s: anchor %r found %d times" % (name, anchor, text.count(anchor)))
    return text.replace(anchor, replacement)


def after_includes(text, prototype):
    """Insert a prototype after the last #include of the leading include block."""
    marker = '#include "MagickCore/'
    last = text.rindex(marker, 0, text.index("\n/*\n", text.index(marker)))
    end = text.index("\n", last) + 1
    return text[:end] + prototype + text[end:]


def morphology(text):
    text = after_includes(text, MORPHOLOGY_PROTOTYPE)
    anchor = "  count = 0;      /* number of low-level morphology primitives performed */\n"
    hook = anchor + '''#if defined(MAGICKCORE_OPENCL_SUPPORT)
  {
    Image *accelerated_image=AccelerateMorphologyApply(image,method,iterations,kernel,
      compose,bias,exception);
    if (accelerated_image != (Image *) NULL)
      return(accelerated_image);
  }
#endif
'''
    return once(text, anchor, hook, "morphology.c")


def effect(text):
    text = once(text, "/* This kernel appears to be broken.\n#if defined(MAGICKCORE_OPENCL_SUPPORT)\n  unsharp_image=AccelerateUnsharpMaskImage(",
                "#if defined(MAGICKCORE_OPENCL_SUPPORT)\n  unsharp_image=AccelerateUnsharpMaskImage(", "effect.c")
    return once(text, "    return(unsharp_image);\n#endif\n*/\n", "    return(unsharp_image);\n#endif\n", "effect.c")


def enhance(text):
    anchor = "  type=IdentifyImageType(image,exception);\n"
    hook = '''#if defined(MAGICKCORE_OPENCL_SUPPORT)
  if (AccelerateContrastStretchImage(image,black_point,white_point,exception) != MagickFalse)
    return(MagickTrue);
#endif
''' + anchor
    begin = text.index("MagickExport MagickBooleanType ContrastStretchImage(")
    end = text.index("MagickExport", begin + 10)
    body = once(text[begin:end], anchor, hook, "enhance.c")
    return text[:begin] + body + text[end:]


def colorspace(text):
    text = after_includes(text, COLORSPACE_PROTOTYPE)
    begin = text.index("MagickExport MagickBooleanType TransformImageColorspace(")
    end = text.index("\n}\n", begin)
    anchor = "  if (colorspace == UndefinedColorspace)\n    return(SetImageColorspace(image,colorspace,exception));\n"
    hook = anchor + '''#if defined(MAGICKCORE_OPENCL_SUPPORT)
  if (AccelerateTransformImageColorspace(image,colorspace,exception) != MagickFalse)
    return(MagickTrue);
#endif
'''
    body = once(text[begin:end], anchor, hook, "colorspace.c")
    return text[:begin] + body + text[end:]


def visual_effects(text):
    # the reference's hook drops `softness`; pass it (the CPU result depends on it)
    text = after_includes(text, WAVELET_PROTOTYPE)
    return once(text, "  noise_image=AccelerateWaveletDenoiseImage(image,threshold,exception);\n",
                "  noise_image=AccelerateWaveletDenoiseImageSoft(image,threshold,softness,exception);\n",
                "visual-effects.c")


def main():
    source, out = sys.argv[1], sys.argv[2]
    os.makedirs(out, exist_ok=True)
    for name, fn in (("morphology.c", morphology), ("effect.c", effect), ("enhance.c", enhance),
                     ("colorspace.c", colorspace), ("visual-effects.c", visual_effects)):
        text = open(os.path.join(source, name), encoding="latin-1").read()
        patched = fn(text)
        with open(os.path.join(out, name), "w", encoding="latin-1") as f:
            f.write('#line 1 "%s"\n' % os.path.join(source, name))
            f.write(patched)


if __name__ == "__main__":
    main()
