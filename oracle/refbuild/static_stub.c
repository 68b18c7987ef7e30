/*
  TEST INFRASTRUCTURE (oracle/_ref build only).

  Stand-in for the reference's MagickCore/static.c, which hard-references the
  Register*Image() entry point of every one of the 142 coders.  The compiled
  reference oracle never reads or writes image files (pixels go in and out
  through the pixel cache), so no coder is registered at all.
  Interface replaced: MagickCore/static.h:25-35.
*/
#include "MagickCore/studio.h"
#include "MagickCore/exception.h"
#include "MagickCore/image.h"
#include "MagickCore/static.h"

MagickExport MagickBooleanType InvokeStaticImageFilter(const char *tag,
  Image **image,const int argc,const char **argv,ExceptionInfo *exception)
{
  (void) tag; (void) image; (void) argc; (void) argv; (void) exception;
  return(MagickFalse);
}

MagickExport MagickBooleanType RegisterStaticModule(const char *module,
  ExceptionInfo *exception)
{
  (void) module; (void) exception;
  return(MagickFalse);
}

MagickExport void RegisterStaticModules(void)
{
}

MagickExport MagickBooleanType UnregisterStaticModule(const char *module)
{
  (void) module;
  return(MagickFalse);
}

MagickExport void UnregisterStaticModules(void)
{
}
